import sys, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, bfutil, blingfire_amd as bf
for wl, model in (("config3", "gpt2.bin"), ("headline512", "gpt2.bin"), ("config3", "roberta.bin")):
    text, off = bfutil.gen_corpus(200000, **bfutil.WORKLOADS[wl]["gen"])
    h = bf.load_model(bfutil.model_path(model)); dt, do = torch.from_numpy(text).cuda(), torch.from_numpy(off).cuda()
    bf.text_to_ids_batch_device(h, dt, do, 2048, 0); torch.cuda.synchronize()
    bf.lib().BfBpeFallbackDocs.restype = ctypes.c_longlong; bf.lib().BfBpeFallbackDocs.argtypes = [ctypes.c_void_p]
    print(wl, model, "fallback docs", bf.lib().BfBpeFallbackDocs(h), "of 200000", bf.last_kernel_ms(h)); bf.free_model(h)
