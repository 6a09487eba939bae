#!/bin/bash
# A/B of kernel variants on the GPU box: tools/gpu_variants.sh "<variants>" [bench args]
vs=$1; shift
mkdir -p gpurun_out
for v in $vs; do
  timeout 300 python bench.py --no-cpu-baseline --variant $v "$@" > gpurun_out/variant_$v.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/variant_$v.log").read().strip().splitlines()[-1])
    print("variant $v: %.1f M docs/s  ms/step %.2f  kernels %s" % (d["value"]/1e6, d["ms_per_step"], {k: round(x,3) for k,x in d["kernel_ms"].items()}))
except Exception as e:
    print("variant $v failed:", e); print(open("gpurun_out/variant_$v.log").read()[-1500:])
PY
done
