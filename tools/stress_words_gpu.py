#!/usr/bin/env python3
"""Stress of the long-document form of the words modes on the GPU: random batches (document lengths 0 .. 6,000 bytes, a few of 50 .. 300 KB, ASCII and
multi-byte text, invalid UTF-8 now and then) through TextToWordsBatch / TextToSentencesBatch under every threshold and test knob (BfSetVariant), compared with
the lane kernel alone (0x40000000; tests/test_words.py pins that one to the reference).  usage: stress_words_gpu.py [rounds] [seed]"""
import ctypes, os, random, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, bfutil, blingfire_amd as bf
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
text, off = bfutil.gen_workload("config1", 10000); raw = text.tobytes()
lines = [raw[off[d]:off[d + 1]] for d in range(10000)]
alpha = ["the", "a", "Dr.", "U.S.", "e-mail", "3,000.50", "д", "好的。", "Ünï", "  ", "\n", "?!", "...", "x" * 40, "　", "don't", "(", ")", "\t"]
def doc():
    k = rnd.random()
    if k < 0.05: return b""
    if k < 0.55: return rnd.choice(lines)[:rnd.randint(1, 200)]
    n = rnd.choice([rnd.randint(1, 300), rnd.randint(300, 6000)]) if k < 0.97 else rnd.randint(50000, 300000)
    out = []; size = 0
    while size < n:
        w = rnd.choice(alpha) if rnd.random() < 0.4 else rnd.choice(lines)[:rnd.randint(1, 60)].decode("utf-8", "ignore")
        out.append(w); size += len(w) + 1
    b = (" " if rnd.random() < 0.8 else "").join(out).encode()[:n]
    if rnd.random() < 0.03: b = b[:len(b) // 2] + b"\xff" + b[len(b) // 2:]
    return b
L = bf.lib(); bad = 0; ncmp = 0
for model, fn in (("wbd.bin", bf.text_to_words_batch), ("sbd.bin", bf.text_to_sentences_batch), ("bert_base_cased_tok.bin", bf.text_to_words_batch)):
    h = bf.load_model(bfutil.model_path(model))
    for r in range(rounds):
        docs = [doc() for _ in range(rnd.randint(1, 400))]
        assert L.BfSetVariant(ctypes.c_void_p(h), 0x40000000) >= 0
        want, woff = fn(docs, h)
        for v in (0, 1 << 12, 2 << 12, 4 << 12, 7 << 12, 0x08000000 | (1 << 12), 0x08000000 | (5 << 12), 0x10000000 | (2 << 12)):
            assert L.BfSetVariant(ctypes.c_void_p(h), v) >= 0
            got, goff = fn(docs, h); ncmp += 1
            if not (np.array_equal(woff, goff) and np.array_equal(want, got)):
                bad += 1
                d = next((i for i in range(len(docs)) if woff[i + 1] != goff[i + 1] or want[woff[i]:woff[i + 1]].tobytes() != got[goff[i]:goff[i + 1]].tobytes()), -1)
                print("MISMATCH", model, "round", r, "variant", hex(v), "first document", d, "of", len(docs), "bytes", len(docs[d]) if d >= 0 else None)
        # the full triple buffer, both ways
        assert L.BfSetVariant(ctypes.c_void_p(h), 0x20000000 | 0x40000000) >= 0
        want, woff = fn(docs, h)
        for v in (0x20000000 | (1 << 12), 0x28000000 | (2 << 12)):
            assert L.BfSetVariant(ctypes.c_void_p(h), v) >= 0
            got, goff = fn(docs, h); ncmp += 1
            if not (np.array_equal(woff, goff) and np.array_equal(want, got)):
                bad += 1; print("MISMATCH (full triple buffer)", model, "round", r, "variant", hex(v))
    bf.free_model(h)
print("stress_words_gpu: %d comparisons, %d mismatches" % (ncmp, bad))
sys.exit(1 if bad else 0)
