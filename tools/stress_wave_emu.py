"""TEST INFRASTRUCTURE: a long randomised run of the WordPiece wave program in the 64-fibre simulator against the oracle -- random batches (empty documents,
long words, runs of one character, raw bytes, the adversarial set), random max_ids / unk / waves / documents per range and simulator configuration (the
TRIM bits of bf_wave_body.h among them).  usage: python tools/stress_wave_emu.py <first seed> <seconds>   (round 4: 6 x 1,200 s, 59,732 batches, all equal)"""
import sys, ctypes, random, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, bfutil, blingfire_amd as bf
import test_wave_emu as T
ht = ctypes.CDLL(bfutil.HOSTTEST_LIB)
ht.bft_load.restype = ctypes.c_void_p; ht.bft_load.argtypes = [ctypes.c_char_p]
ht.bft_emu_wave_batch.restype = ctypes.c_long
ht.bft_emu_wave_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
seed0 = int(sys.argv[1]); budget = float(sys.argv[2])
ora = bfutil.oracle()
models = [m for m in ("bert_base_tok.bin", "bert_base_cased_tok.bin", "bert_chinese.bin") if bfutil.have_model(m)]
hs = {m: (ht.bft_load(bfutil.model_path(m).encode()), ora.load(bfutil.model_path(m))) for m in models}
t0 = time.time(); n = 0; seed = seed0
alpha = "abcdefghijklmnopqrstuvwxyz"
while time.time() - t0 < budget:
    rnd = random.Random(seed); seed += 1
    docs = []
    kind = rnd.randrange(6)
    nd = rnd.randint(1, 60)
    for _ in range(nd):
        k = rnd.randrange(8)
        if k == 0: docs.append(b"")
        elif k == 1: docs.append(("".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 1200)))).encode())
        elif k == 2: docs.append((" ".join("".join(rnd.choice(alpha + "A.,é好") for _ in range(rnd.randint(1, 14))) for _ in range(rnd.randint(1, 200)))).encode())
        elif k == 3: docs.append(bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 80))))
        elif k == 4: docs.append((rnd.choice(["a", ".", " ", "好", "é", "##ing", "[UNK]"]) * rnd.randint(1, 700)).encode())
        elif k == 5: docs.append(rnd.choice(bfutil.ADVERSARIAL))
        elif k == 6: docs.append(("x" * rnd.randint(500, 530) + " y").encode())
        else: docs.append((" ".join(rnd.choice(["the", "unaffable", "telescope", "of", "a", "internationalization"]) for _ in range(rnd.randint(1, 120)))).encode())
    text, off = bf.pack_docs(docs)
    m = rnd.choice(models); h, ho = hs[m]
    mx = rnd.choice([0, 1, 3, 64, 512, 1 << 20]); unk = rnd.choice([0, 100, 7]); nw = rnd.randint(1, 5); grab = rnd.randint(1, 8)
    cfg = rnd.choice([3, 3, 4, 5, 19, 20, 0])
    r, ids, ido, _ = T.wave_batch(ht, h, text, off, mx, unk, nw, grab, cfg)
    gids, goff = ora.batch(ho, text, off, mx, unk)
    assert r >= 0 and np.array_equal(ido, goff) and np.array_equal(ids, gids), (seed - 1, m, mx, unk, nw, grab, cfg)
    n += 1
print("seed0", seed0, "batches", n, "ok")
