#!/usr/bin/env python3
"""BASELINE.json configs[0] (default pattern tokenizer, TextToWords, short English lines) on the GPU batch entry point, text
resident in HBM, next to the reference's per-line CPU call (one thread, ctypes loop: call overhead included)."""
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
if bfutil.have_ref():
    ref = bfutil.reference(); g = ref.lib.TextToWords; g.restype = ctypes.c_int; g.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    raw = text.tobytes(); o = ctypes.create_string_buffer(1024); n = min(nd, 100000)
    t0 = time.perf_counter()
    for d in range(n): g(raw[off[d]:off[d + 1]], int(off[d + 1] - off[d]), o, 1024)
    t = time.perf_counter() - t0
    print("reference TextToWords, 1 thread, %d lines: %.2f M lines/s" % (n, n / t / 1e6))
