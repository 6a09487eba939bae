#!/bin/bash
# quick A/B of kernel variants: tools/gpu_ab.sh "<variant> <variant> ..." [bench args]   (-1 = product default)
set -u
O=$PWD/gpurun_out/ab; mkdir -p $O
Q="--no-cpu-baseline --no-extra-timings --verify 0 --steps 5 --warmup 2 ${2:-}"
for v in $1; do
  timeout 300 python bench.py $Q --variant $v > $O/v_$v.json 2> $O/v_$v.err
  python - $O/v_$v.json $v <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print("variant", sys.argv[2], "ms/step %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernel_ms"].items()})
PY
done
