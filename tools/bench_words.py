#!/usr/bin/env python3
"""BASELINE.json configs[0] (default pattern tokenizer, TextToWords, short English lines) on the GPU batch entry point, text
resident in HBM.  usage: bench_words.py [lines] [variant] [config1]: a variant (BfSetVariant: 0x40000000 = no long-document path,
k << 12 = documents of more than 8 << k characters take it) runs models/wbd.bin as a loaded model instead of the built-in one;
`config1` = the reference's own 10,000 lines instead of the synthetic corpus."""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, bfutil, blingfire_amd as bf
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
variant = int(sys.argv[2], 0) if len(sys.argv) > 2 else None
if len(sys.argv) > 3 and sys.argv[3] == "config1": text, off = bfutil.gen_workload("config1", nd)
else: text, off = bfutil.gen_corpus(nd, seed=1, mean=43, sd=12, minlen=8, maxlen=120)
L = bf.lib(); dt, do = torch.from_numpy(text).cuda(), torch.from_numpy(off).cuda()
h = None
if variant is not None:
    h = bf.load_model(bfutil.model_path("wbd.bin")); assert L.BfSetVariant(ctypes.c_void_p(h), variant) >= 0
out = torch.empty(2 * len(text) + 64, dtype=torch.uint8, device="cuda"); t_off = torch.empty(nd + 1, dtype=torch.int64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def step():
    r = L.TextToWordsBatchDevice(ctypes.c_void_p(h) if h else None, dt.data_ptr(), do.data_ptr(), nd, len(text), out.data_ptr(), out.numel(), t_off.data_ptr(), ctypes.c_void_p(s))
    assert r == 0, r
for _ in range(3): step()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): step()
e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 10
print("TextToWordsBatchDevice (%s): %d lines, %.1f B/line: %.3f ms/step, %.1f M lines/s, %.2f GB/s of text in, %d bytes out"
      % ("built-in wbd.bin" if h is None else "wbd.bin, variant 0x%x" % variant, nd, len(text) / nd, ms, nd / ms / 1e3, len(text) / ms / 1e6, int(t_off[-1].item())))
