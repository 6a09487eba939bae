"""TEST INFRASTRUCTURE: a long randomised run of the WordPiece FLAT program (bf_flat_body.h: k_wp_flat, k_wp_units, the LIST wave program, k_wp_count, k_wp_merge
from the device sources) in the 64-fibre simulator against the oracle -- random batches (empty documents, long words, runs of one character, raw bytes,
many-piece words, the adversarial set), random max_ids / unk / waves / ranges; ids and, every other batch, the byte offsets of every id.
usage: python tools/stress_flat_emu.py <first seed> <seconds>   (round 5: 12,000 random batches of up to 90 documents, half of them with offsets, found one
defect -- a chunk of 512 one-byte tokens behind a run that ended with the chunk before it puts 513 tokens on a list that had 512 places: the process dies, the
last seed is the batch -- and 20,633 batches after the fix were all equal)"""
import sys, ctypes, random, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, bfutil, blingfire_amd as bf
import test_flat_emu as T
ht = ctypes.CDLL(bfutil.HOSTTEST_LIB)
ht.bft_load.restype = ctypes.c_void_p; ht.bft_load.argtypes = [ctypes.c_char_p]
ht.bft_emu_flat_batch.restype = ctypes.c_long
ht.bft_emu_flat_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
ht.bft_emu_flat_batch_offsets.restype = ctypes.c_long
ht.bft_emu_flat_batch_offsets.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
seed0 = int(sys.argv[1]); budget = float(sys.argv[2])
ora = bfutil.oracle()
models = [m for m in ("bert_base_tok.bin", "bert_base_cased_tok.bin", "bert_chinese.bin", "bert_multi_cased_tok.bin") if bfutil.have_model(m)]
hs = {m: (ht.bft_load(bfutil.model_path(m).encode()), ora.load(bfutil.model_path(m))) for m in models}
t0 = time.time(); n = 0; noff = 0; seed = seed0
alpha = "abcdefghijklmnopqrstuvwxyz"
while time.time() - t0 < budget:
    rnd = random.Random(seed); seed += 1
    docs = []
    nd = rnd.randint(1, 90)
    for _ in range(nd):
        k = rnd.randrange(10)
        if k == 0: docs.append(b"")
        elif k == 1: docs.append(("".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 1200)))).encode())
        elif k == 2: docs.append((" ".join("".join(rnd.choice(alpha + "A.,é好") for _ in range(rnd.randint(1, 14))) for _ in range(rnd.randint(1, 200)))).encode())
        elif k == 3: docs.append(bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 80))))
        elif k == 4: docs.append((rnd.choice(["a", ".", " ", "好", "é", "##ing", "[UNK]", "ab ", "x, "]) * rnd.randint(1, 700)).encode())
        elif k == 5: docs.append(rnd.choice(bfutil.ADVERSARIAL))
        elif k == 6: docs.append(("x" * rnd.randint(40, 60) + " y").encode())
        elif k == 7: docs.append((" ".join("".join(rnd.choice("qzxjkvw") for _ in range(rnd.randint(6, 16))) for _ in range(rnd.randint(1, 300)))).encode())
        elif k == 8: docs.append((" ".join(rnd.choice(["the", "unaffable", "telescope", "of", "a", "internationalization", "café", "naïve", "3,000.50", "e-mail"]) for _ in range(rnd.randint(1, 160)))).encode())
        else: docs.append(("".join(rnd.choice([" ", "a", "b", "é", "一", ".", "\U00020000", "﻿"]) for _ in range(rnd.randint(1, 40)))).encode())
    text, off = bf.pack_docs(docs)
    m = rnd.choice(models); h, ho = hs[m]
    mx = rnd.choice([0, 1, 3, 64, 512, 1 << 20]); unk = rnd.choice([0, 100, 7]); nw = rnd.randint(1, 5); nr = rnd.choice([0, 0, 1, 2, 3, 7, 40])
    gids, goff = ora.batch(ho, text, off, mx, unk)
    if seed & 1:
        r, ids, ido, _ = T.flat_batch(ht, h, text, off, mx, unk, nw, nr)
        assert r >= 0 and np.array_equal(ido, goff) and np.array_equal(ids, gids), (seed - 1, m, mx, unk, nw, nr, r)
    else:
        r, ids, sts, ens, ido, _ = T.flat_batch_offsets(ht, h, text, off, mx, unk, nw, nr)
        assert r >= 0 and np.array_equal(ido, goff) and np.array_equal(ids, gids), (seed - 1, m, mx, unk, nw, nr, r, "offsets form")
        raw = text.tobytes(); ws, we = [], []
        for d in range(len(off) - 1):
            c, _, s_, e_ = ora.with_offsets(ho, raw[off[d]:off[d + 1]], mx, unk, "bfo_text_to_ids_with_offsets")
            ws += s_[:min(c, mx)]; we += e_[:min(c, mx)]
        assert sts.tolist() == ws and ens.tolist() == we, (seed - 1, m, mx, unk, nw, nr, "offsets")
        noff += 1
    n += 1
print("seed0", seed0, "batches", n, "with offsets", noff, "ok")
