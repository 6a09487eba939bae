"""Counters of k_uni_walk on a sample of configs 4 / 5 (BF_LEX_STATS=1): entries per start, documents flagged for the stage: tools/uni_walk_stats.py [ndocs]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["BF_LEX_STATS"] = "1"
import numpy as np, torch, bfutil, blingfire_amd as bf
ndocs = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
for wl in ("config4", "config5"):
    w = bfutil.WORKLOADS[wl]
    text, off = bfutil.gen_workload(wl, ndocs)
    h = bf.load_model(bfutil.model_path(w["model"]))
    for v in (7, 8):
        bf.lib().BfSetVariant(h, v)
        bf.lib().BfSetLexStats.argtypes = [ctypes.c_void_p, ctypes.c_int]; bf.lib().BfSetLexStats(h, 1)
        d_text = torch.from_numpy(text.copy()).cuda(); d_off = torch.from_numpy(off.copy()).cuda()
        bf.text_to_ids_batch_device(h, d_text, d_off, w["max_ids"], w["unk"]); torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 16)()
        bf.lib().BfLexStats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        bf.lib().BfLexStats(h, out, 16)
        st = [int(out[i]) for i in range(16)]
        print(wl, "variant", v, "docs", ndocs, "rounds", st[0], "steps", st[1], "records", st[2], "transitions", st[3], "pieces", st[4], "docs flagged for the stage", st[5], "| starts with 0..7+ entries", st[8:16], flush=True)
    bf.free_model(h)
