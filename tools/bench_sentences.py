#!/usr/bin/env python3
"""TextToSentencesBatchDevice (built-in sbd.bin) on text resident in HBM: one long document against the same bytes as short documents.
usage: bench_sentences.py [bytes of the long document] [variant] [case 0..2, -1 = all] [noruns]: a variant (BfSetVariant: 0x40000000 = no long-document path, k << 12 = documents of
more than 8 << k characters take it) runs models/sbd.bin as a loaded model instead of the built-in one"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, bfutil, blingfire_amd as bf
nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
text, off = bfutil.gen_workload("config1", 10000)
raw = text.tobytes(); lines = [raw[off[d]:off[d + 1]] for d in range(10000)]
if len(sys.argv) > 4 and sys.argv[4] == "noruns":      # without the lines that hold a run of ten spaces (one start position of such a line costs sbd.bin 6,440 sequential steps)
    lines = [l for l in lines if b" " * 10 not in l]; print("(%d lines without a run of ten spaces)" % len(lines))
big = (b". ".join(lines * (nbytes // len(raw) + 1)))[:nbytes]
L = bf.lib(); s = torch.cuda.current_stream().cuda_stream
variant = int(sys.argv[2], 0) if len(sys.argv) > 2 else None
h = None
if variant is not None:
    h = bf.load_model(bfutil.model_path("sbd.bin")); assert L.BfSetVariant(ctypes.c_void_p(h), variant) >= 0; print("sbd.bin, variant 0x%x" % variant)
def run(docs, label):
    t, o = bf.pack_docs(docs); t = t.copy(); o = o.copy()
    dt, do = torch.from_numpy(t).cuda(), torch.from_numpy(o).cuda(); nd = len(docs)
    out = torch.empty(2 * len(t) + 64, dtype=torch.uint8, device="cuda"); t_off = torch.empty(nd + 1, dtype=torch.int64, device="cuda")
    def step():
        r = L.TextToSentencesBatchDevice(ctypes.c_void_p(h) if h else None, dt.data_ptr(), do.data_ptr(), nd, len(t), out.data_ptr(), out.numel(), t_off.data_ptr(), ctypes.c_void_p(s))
        assert r == 0, r
    for _ in range(3): step()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 10
    print("TextToSentencesBatchDevice, %s: %d documents, %d bytes: %.3f ms/step, %.1f MB/s, %d bytes out" % (label, nd, len(t), ms, len(t) / ms / 1e3, int(t_off[-1].item())))
case = int(sys.argv[3]) if len(sys.argv) > 3 else -1
if case in (-1, 0): run([big], "one document")
if case in (-1, 1): run([big[i * 100:(i + 1) * 100] for i in range(max(1, nbytes // 100))], "the same bytes as 100-byte documents")
if case in (-1, 2): run([big[i * 1000:(i + 1) * 1000] for i in range(max(1, nbytes // 1000))], "the same bytes as 1000-byte documents")
