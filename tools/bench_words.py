#!/usr/bin/env python3
"""BASELINE.json configs[0] (default pattern tokenizer, TextToWords, short English lines) on the GPU batch entry point, text
resident in HBM."""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, bfutil, blingfire_amd as bf
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
text, off = bfutil.gen_corpus(nd, seed=1, mean=43, sd=12, minlen=8, maxlen=120)
L = bf.lib(); dt, do = torch.from_numpy(text).cuda(), torch.from_numpy(off).cuda()
out = torch.empty(2 * len(text) + 64, dtype=torch.uint8, device="cuda"); t_off = torch.empty(nd + 1, dtype=torch.int64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def step():
    r = L.TextToWordsBatchDevice(None, dt.data_ptr(), do.data_ptr(), nd, len(text), out.data_ptr(), out.numel(), t_off.data_ptr(), ctypes.c_void_p(s))
    assert r == 0, r
for _ in range(2): step()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): step()
e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
print("TextToWordsBatchDevice (built-in wbd.bin): %d lines, %.1f B/line: %.3f ms/step, %.1f M lines/s, %.2f GB/s of text in, %d bytes out"
      % (nd, len(text) / nd, ms, nd / ms / 1e3, len(text) / ms / 1e6, int(t_off[-1].item())))
