"""Wall time of TextToIdsBatch on batches that hold one long run of a character (gpt2.bin / roberta.bin): tools/bpe_long_runs.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, bfutil, blingfire_amd as bf
for model in ("gpt2.bin", "roberta.bin"):
    h = bf.load_model(bfutil.model_path(model))
    base = bfutil.fuzz_docs(200, seed=1)
    bf.text_to_ids_batch(h, bf.pack_docs(base), 1 << 22, 0)
    for ch in (b"-", b"="):
        for n in (100, 1500, 10000, 50000, 200000, 1000000):
            text, off = bf.pack_docs(base + [ch * n])
            best = 1e9
            for rep in range(2):
                t0 = time.time(); ids, ido = bf.text_to_ids_batch(h, (text, off), 1 << 22, 0); best = min(best, time.time() - t0)
            print("%s  %r x %7d: %8.2f ms (best of 2; the first pass of a size may grow the pool), %d ids for the run" % (model, ch, n, best * 1e3, int(ido[-1] - ido[-2])), flush=True)
    bf.free_model(h)
