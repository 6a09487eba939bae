#!/usr/bin/env python3
"""Throughput of the batch IdsToText path (tokenise a corpus with gpt2.bin, detokenise the ids with gpt2.i2w), ids resident in HBM."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, bfutil, blingfire_amd as bf
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
text, off = bfutil.gen_corpus(nd, **bfutil.WORKLOADS["config3"]["gen"])
ht, hd = bf.load_model(bfutil.model_path("gpt2.bin")), bf.load_model(bfutil.model_path("gpt2.i2w"))
dt, do = torch.from_numpy(text).cuda(), torch.from_numpy(off).cuda()
ids, id_off = bf.text_to_ids_batch_device(ht, dt, do, 2048, 0)
torch.cuda.synchronize()
n_ids = int(id_off[-1].item()); ids = ids[:n_ids].contiguous()
out = torch.empty(int(off[-1]) + 64, dtype=torch.uint8, device="cuda"); t_off = torch.empty(nd + 1, dtype=torch.int64, device="cuda")
for _ in range(2): bf.ids_to_text_batch_device(hd, ids, id_off, out, t_off, False)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): bf.ids_to_text_batch_device(hd, ids, id_off, out, t_off, False)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5; nbytes = int(t_off[-1].item())
alg = 2 * 4 * n_ids + 2 * nbytes + 2 * 16 * nd          # both passes read the ids; token bytes read + written; offsets
print("IdsToTextBatchDevice: %d sequences, %d ids, %d text bytes: %.3f ms/step, %.1f M seq/s, %.2f G ids/s, %.1f GB/s of text out, algorithmic %.0f GB/s (%.1f%% of 8 TB/s)"
      % (nd, n_ids, nbytes, ms, nd / ms / 1e3, n_ids / ms / 1e6, nbytes / ms / 1e6, alg / ms / 1e6, 100 * alg / ms / 1e6 / 8000))
