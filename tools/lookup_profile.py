#!/usr/bin/env python3
"""Where do the lexer's table lookups land?  Runs the lane program on the host (tests/hosttest) over a sample of the
headline corpus and prints the cumulative share of lookups below each table index -- the evidence for keeping a
prefix of the table in LDS."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, bfutil
ndocs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
model = sys.argv[2] if len(sys.argv) > 2 else bfutil.bert_model_name()
wl = sys.argv[3] if len(sys.argv) > 3 else "headline512"
L = ctypes.CDLL(bfutil.HOSTTEST_LIB)
L.bft_load.restype = ctypes.c_void_p; L.bft_load.argtypes = [ctypes.c_char_p]
L.bft_table_len.restype = ctypes.c_long; L.bft_table_len.argtypes = [ctypes.c_void_p]
L.bft_emu_text_to_ids.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
h = L.bft_load(bfutil.model_path(model).encode())
text, off = bfutil.gen_corpus(ndocs, **bfutil.WORKLOADS[wl]["gen"])
raw = text.tobytes(); ids = (ctypes.c_int32 * 4096)()
hist = (ctypes.c_ulonglong * 4096)()
L.bft_lookup_hist(hist, 4096, 1)
for d in range(ndocs):
    b = raw[off[d]:off[d + 1]]
    L.bft_emu_text_to_ids(h, b, len(b), ids, 512, 100)
L.bft_lookup_hist(hist, 4096, 0)
a = np.array(hist[:], dtype=np.float64); tot = a.sum(); c = np.cumsum(a) / tot
print("model", model, "table entries", L.bft_table_len(h), "lookups", int(tot), "per doc %.0f" % (tot / ndocs))
for n in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 128, 256):
    print("  idx < %6d (%4d KB at 8 B, %4d KB at 4 B): %.1f%%" % (n * 1024, n * 8, n * 4, 100 * c[min(n * 4, 4095) - 1]))
