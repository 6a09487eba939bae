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
trips, walk_lanes, ev_rounds, ev_lanes, fetch_rounds, need_lanes = [out[i] for i in range(6)]
print("docs", ndocs, "wave-trips", trips, "lane-steps", walk_lanes, "walk lane efficiency %.1f%%" % (100.0 * walk_lanes / (64.0 * trips)),
      "idle(need) %.1f%%" % (100.0 * need_lanes / (64.0 * trips)))
print("event rounds", ev_rounds, "lanes/round %.1f" % (ev_lanes / max(ev_rounds, 1)), "events per doc %.1f" % (ev_lanes / ndocs),
      "steps per doc %.1f" % (walk_lanes / ndocs), "trips per event round %.2f" % (trips / max(ev_rounds, 1)), "fetch rounds", fetch_rounds)
tw, te, tf = out[6], out[7], out[8]
tot = tw + te + tf
print("wave-cycles: walk %.1f%% event %.1f%% fetch %.1f%%;  cycles per trip %.0f, per event round %.0f, per fetch round %.0f" % (
      100.0*tw/tot, 100.0*te/tot, 100.0*tf/tot, tw/max(trips,1), te/max(ev_rounds,1), tf/max(fetch_rounds,1)))
