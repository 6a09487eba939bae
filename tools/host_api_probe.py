"""Where the wall time of TextToIdsBatch on host buffers goes (BF_TRACE_HOST=1 prints the library's own split): the default workload,
1.25 M documents, output arrays allocated once and touched before the timed calls."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bfutil
import blingfire_amd as bf

os.environ["BF_TRACE_HOST"] = "1"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
text, off = bfutil.gen_workload("headline512", n)
h = bf.load_model(bfutil.model_path(bfutil.bert_model_name()))
L = bf.lib()
cap = int(off[-1]) + 1
ids = np.zeros(cap, dtype=np.int32)
ioff = np.zeros(n + 1, dtype=np.int64)
for chunk in (128 << 20, 256 << 20):
    L.BfSetHostChunkBytes.restype = ctypes.c_int64
    L.BfSetHostChunkBytes.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.BfSetHostChunkBytes(ctypes.c_void_p(h), chunk)
    for it in range(3):
        t = time.perf_counter()
        r = L.TextToIdsBatch(ctypes.c_void_p(h), text.ctypes.data, off.ctypes.data, n, ids.ctypes.data, cap, ioff.ctypes.data, 512, 100)
        dt = time.perf_counter() - t
        print("chunk %d MiB, call %d: %.1f ms, %.1f M docs/s, %d ids" % (chunk >> 20, it, dt * 1e3, n / dt / 1e6, r), flush=True)

# ---- the same batch range-sharded over two logical shards of the one device (BfSetDevices): throughput against the one-handle call above,
# and what the call adds to the resident set (the caller's input and output arrays are touched already)
import resource
def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
L.BfSetHostChunkBytes(ctypes.c_void_p(h), 128 << 20)
one = ids[:max(r, 0)].copy() if r > 0 else None
base_rss = rss_mb()
for G in (2, 4):
    bf.set_devices(h, [0] * G)
    for it in range(3):
        t = time.perf_counter()
        r2 = L.TextToIdsBatch(ctypes.c_void_p(h), text.ctypes.data, off.ctypes.data, n, ids.ctypes.data, cap, ioff.ctypes.data, 512, 100)
        dt = time.perf_counter() - t
        same = one is not None and r2 == len(one) and np.array_equal(ids[:r2], one)
        print("BfSetDevices G = %d on one device, call %d: %.1f ms, %.1f M docs/s, %d ids, equal to G = 1: %s, peak RSS %.0f MB (before the sharded calls %.0f MB; input %.0f MB, output arrays %.0f MB)"
              % (G, it, dt * 1e3, n / dt / 1e6, r2, same, rss_mb(), base_rss, len(text) / 1e6, (ids.nbytes + ioff.nbytes) / 1e6), flush=True)
bf.set_devices(h, [0])
