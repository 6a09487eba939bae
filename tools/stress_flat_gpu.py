"""TEST INFRASTRUCTURE: random batches through the FLAT program on the GPU (C-ABI, BfSetVariant 4) against the CPU checker -- the generator of
tools/stress_flat_emu.py at batch sizes the device is meant for (hundreds to thousands of documents, so that ranges, chunk alignments and document
boundaries fall everywhere), ids of every batch, byte offsets of every id of every fourth.  usage: python tools/stress_flat_gpu.py <first seed> <seconds>   (round 5: two runs of 5 minutes on an MI355X, 1,052 batches / 1,461,600 documents / 197 batches with offsets, all equal)"""
import sys, random, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, bfutil, blingfire_amd as bf
seed0 = int(sys.argv[1]); budget = float(sys.argv[2])
ck = bfutil.reference() if bfutil.have_ref() else bfutil.oracle()
name = "TextToIdsWithOffsets" if bfutil.have_ref() else "bfo_text_to_ids_with_offsets"
models = [m for m in ("bert_base_tok.bin", "bert_base_cased_tok.bin", "bert_chinese.bin", "bert_multi_cased_tok.bin") if bfutil.have_model(m)]
hs = {m: (bf.load_model(bfutil.model_path(m)), ck.load(bfutil.model_path(m))) for m in models}
for m in models: bf.lib().BfSetVariant(hs[m][0], 4)
alpha = "abcdefghijklmnopqrstuvwxyz"
t0 = time.time(); n = 0; noff = 0; ndocs = 0; seed = seed0
while time.time() - t0 < budget:
    rnd = random.Random(seed); seed += 1
    docs = []
    for _ in range(rnd.choice([40, 300, 1500, 4000])):
        k = rnd.randrange(11)
        if k == 0: docs.append(b"")
        elif k == 1: docs.append(("".join(rnd.choice(alpha) for _ in range(rnd.randint(1, 1200)))).encode())
        elif k == 2: docs.append((" ".join("".join(rnd.choice(alpha + "A.,é好") for _ in range(rnd.randint(1, 14))) for _ in range(rnd.randint(1, 200)))).encode())
        elif k == 3: docs.append(bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 80))))
        elif k == 4: docs.append((rnd.choice(["a", ".", " ", "好", "é", "##ing", "[UNK]", "ab ", "x, "]) * rnd.randint(1, 700)).encode())
        elif k == 5: docs.append(rnd.choice(bfutil.ADVERSARIAL))
        elif k == 6: docs.append(("x" * rnd.randint(40, 60) + " y").encode())
        elif k == 7: docs.append((" ".join("".join(rnd.choice("qzxjkvw") for _ in range(rnd.randint(6, 16))) for _ in range(rnd.randint(1, 300)))).encode())
        elif k == 8: docs.append((" ".join(rnd.choice(["the", "unaffable", "telescope", "of", "a", "internationalization", "café", "naïve", "3,000.50", "e-mail"]) for _ in range(rnd.randint(1, 160)))).encode())
        elif k == 9: docs.append(("".join(rnd.choice([" ", "a", "b", "é", "一", ".", "\U00020000", "﻿"]) for _ in range(rnd.randint(1, 40)))).encode())
        else: docs.append((b"some words and " * 40)[:rnd.randint(480, 530)] + rnd.choice([b"", b"zqxjkvw", b"." * rnd.randint(500, 1100)]))
    text, off = bf.pack_docs(docs)
    m = rnd.choice(models); h, hck = hs[m]
    mx = rnd.choice([1, 3, 64, 512, 1 << 20]); unk = rnd.choice([0, 100, 7])
    want_ids, want_off = ck.batch(hck, text, off, mx, unk)
    ids, id_off = bf.text_to_ids_batch(h, (text, off), mx, unk)
    assert np.array_equal(id_off, want_off) and np.array_equal(ids, want_ids), (seed - 1, m, mx, unk, len(docs))
    if seed % 4 == 0 and len(docs) <= 1500:
        gi, gs, ge, go = bf.text_to_ids_with_offsets_batch(h, docs, mx, unk)
        assert np.array_equal(go, want_off) and np.array_equal(gi, want_ids), (seed - 1, m, mx, unk, "offsets form: ids")
        for d in rnd.sample(range(len(docs)), min(len(docs), 200)):
            c, wi, ws, we = ck.with_offsets(hck, docs[d], mx, unk, name)
            c = min(c, mx)
            assert gs[go[d]:go[d + 1]].tolist() == ws[:c] and ge[go[d]:go[d + 1]].tolist() == we[:c], (seed - 1, m, mx, unk, d, "offsets")
        noff += 1
    n += 1; ndocs += len(docs)
print("seed0", seed0, "batches", n, "documents", ndocs, "with offsets", noff, "ok")
