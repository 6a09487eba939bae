#!/bin/bash
# wave program against the lane-per-document kernels (variant 2) on the other WordPiece shapes: config 2 (128-byte documents), one 200 KB document
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/wave_vs_lane; mkdir -p $O
Q="--no-cpu-baseline --no-extra-timings --steps 5 --warmup 2"
for v in 3 2; do
  timeout 300 python bench.py $Q --workload config2 --variant $v > $O/config2_v$v.json 2> $O/config2_v$v.err
  python - $O/config2_v$v.json $v <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print("config2 variant", sys.argv[2], "verified", j.get("verified_docs"), "ms/step %.3f" % j["ms_per_step"], {k: round(v, 3) for k, v in j["kernel_ms"].items()})
PY
done
timeout 300 python - <<'PY'
import time, numpy as np, torch, sys
sys.path.insert(0, "tests")
import bfutil, blingfire_amd as bf
rnd = np.random.RandomState(1)
words = [bytes(rnd.randint(97, 123, size=rnd.randint(1, 12)).astype(np.uint8)) for _ in range(40000)]
doc = b" ".join(words)
print("document bytes", len(doc))
for variant in (3, 2):
    h = bf.load_model(bfutil.model_path(bfutil.bert_model_name()))
    bf.lib().BfSetVariant(h, variant)
    text, off = bf.pack_docs([doc] + [b"hello world"] * 1000)
    d_text = torch.from_numpy(text).cuda(); d_off = torch.from_numpy(off).cuda()
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ids, ido = bf.text_to_ids_batch_device(h, d_text, d_off, 1 << 20, 100)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("variant", variant, "one 200 KB document + 1000 short ones: %.2f ms, ids %d" % (dt * 1e3, int(ido[-1])))
    bf.free_model(h)
PY
