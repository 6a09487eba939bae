"""Counters of the BPE wave program on the config-3 corpus (BF_LEX_STATS=1): tools/bpe_wave_stats.py [ndocs]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["BF_LEX_STATS"] = "1"
import numpy as np, torch, bfutil, blingfire_amd as bf
ndocs = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
text, off = bfutil.gen_workload("config3", ndocs)
h = bf.load_model(bfutil.model_path("gpt2.bin"))
d_text = torch.from_numpy(text).cuda(); d_off = torch.from_numpy(off).cuda()
bf.text_to_ids_batch_device(h, d_text, d_off, 2048, 0); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
bf.lib().BfLexStats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
bf.lib().BfLexStats(h, out, 16)
st = [int(out[i]) for i in range(16)]
print("docs", ndocs, "words", st[0], "taken whole", st[1], "handed back: symbol", st[2], "word too long", st[3], "window", st[4], "start without arc", st[5], "no applied arc", st[6],
      "| solved words by arcs <=16/24/32/48/more", st[8:13], "| documents flagged", st[13], "of which with text", st[14])
