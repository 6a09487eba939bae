#!/usr/bin/env python3
"""TextToIdsBatch on pageable host arrays (the chunked host path), 5 M documents of the metric's corpus, three calls: wall clock per call.
Run it under `taskset -c <cpus of one NUMA node>` to see what the node of the caller's threads and arrays does to the figure
(tools/gpu_r6_numa.sh).  usage: host_numa_probe.py [documents]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bfutil
import blingfire_amd as bf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
text, off = bfutil.gen_workload("headline512", n)
h = bf.load_model(bfutil.model_path(bfutil.bert_model_name()))
L = bf.lib()
cap = int(off[-1]) + 1
ids = np.zeros(cap, dtype=np.int32); ioff = np.zeros(n + 1, dtype=np.int64)
best = 0
for it in range(4):
    t = time.perf_counter()
    r = L.TextToIdsBatch(ctypes.c_void_p(h), text.ctypes.data, off.ctypes.data, n, ids.ctypes.data, cap, ioff.ctypes.data, 512, 100)
    dt = time.perf_counter() - t
    if it: best = max(best, n / dt / 1e6)
print("cpus %s: TextToIdsBatch, %d documents: best of 3 %.1f M docs/s (%d ids)" % (os.sched_getaffinity(0) and ("%d..%d (%d)" % (min(os.sched_getaffinity(0)), max(os.sched_getaffinity(0)), len(os.sched_getaffinity(0)))), n, best, r), flush=True)
try:
    roll = open("/proc/self/smaps_rollup").read()
    print("   " + " ".join(l.strip() for l in roll.splitlines() if l.startswith(("Rss", "AnonHugePages"))), "| THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), flush=True)
except Exception as e:
    print("   (no smaps_rollup: %s)" % e)
