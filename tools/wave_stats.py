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
win, slow, flush, tok, trips, steps, rewalk = [out[i] for i in range(7)]
print("docs", ndocs, "bytes", int(off[-1]), "windows/doc %.2f" % (win / ndocs), "slow windows %.4f" % (slow / max(win, 1)), "flushes/doc %.2f" % (flush / ndocs),
      "tokens/doc %.1f" % (tok / ndocs), "tokens/flush %.1f" % (tok / max(flush, 1)), "phase-B trips/flush %.2f" % (trips / max(flush, 1)),
      "active slots per trip %.1f of 128" % (steps / max(trips, 1)), "rewalks/doc %.3f" % (rewalk / ndocs))
