#!/usr/bin/env python3
"""Single-document TextToIds through the drop-in entry point: calls/s from 1 and from T host threads on ONE handle (concurrent
callers are combined into shared launches, bf_capi.cpp text_to_ids_one), next to the reference CPU library called the same way."""
import ctypes, os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, bfutil, blingfire_amd as bf
ndocs = 4000
text, off = bfutil.gen_workload("config2", ndocs)
raw = text.tobytes(); docs = [raw[off[i]:off[i + 1]] for i in range(ndocs)]
model = bfutil.model_path(bfutil.bert_model_name())

def run(call, h, T):
    def work(t):
        buf = (ctypes.c_int32 * 512)()
        for k in range(t, ndocs, T):
            call(ctypes.c_void_p(h), docs[k], len(docs[k]), buf, 512, 100)
    ts = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    return ndocs / (time.perf_counter() - t0)

L = bf.lib(); h = bf.load_model(model)
for T in (1, 4, 16, 64):
    run(L.TextToIds, h, T)
    print("GPU library   %3d threads: %9.0f calls/s" % (T, run(L.TextToIds, h, T)), flush=True)
bf.free_model(h)
if bfutil.have_ref():
    Rf = ctypes.CDLL(bfutil.REF_LIB); Rf.LoadModel.restype = ctypes.c_void_p; Rf.LoadModel.argtypes = [ctypes.c_char_p]
    Rf.TextToIds.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    hr = Rf.LoadModel(model.encode())
    for T in (1, 4, 16):
        print("reference CPU %3d threads: %9.0f calls/s (ctypes releases the GIL inside the call)" % (T, run(Rf.TextToIds, hr, T)), flush=True)
