"""Counters of the wave kernel (bf_wave.h) on the headline corpus: tools/wave_stats.py [ndocs] [variant].  BF_LEX_STATS instances."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["BF_LEX_STATS"] = "1"
import numpy as np, torch, bfutil, blingfire_amd as bf
ndocs = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
variant = int(sys.argv[2]) if len(sys.argv) > 2 else -1
wl = bfutil.WORKLOADS["headline512"]
text, off = bfutil.gen_corpus(ndocs, **wl["gen"])
h = bf.load_model(bfutil.model_path(bfutil.bert_model_name()))
if variant >= 0: bf.lib().BfSetVariant(h, variant)
d_text = torch.from_numpy(text).cuda(); d_off = torch.from_numpy(off).cuda()
bf.text_to_ids_batch_device(h, d_text, d_off, 512, 100); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
bf.lib().BfLexStats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
bf.lib().BfLexStats(h, out, 16)
st = [int(out[i]) for i in range(16)]
print("docs", ndocs, "bytes", int(off[-1]), "unit trips/doc %.1f fast windows/doc %.2f general windows/doc %.3f tokens/doc %.1f unit-steps/doc %.0f (per trip %.1f) retires/doc %.2f rewalks/doc %.3f "
      "trips without progress/doc %.3f decodes/doc %.2f" % (st[0] / ndocs, st[1] / ndocs, st[2] / ndocs, st[3] / ndocs, st[4] / ndocs, st[4] / max(st[0], 1), st[5] / ndocs, st[6] / ndocs, st[7] / ndocs, st[8] / ndocs))
