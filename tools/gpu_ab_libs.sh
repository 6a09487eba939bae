#!/bin/bash
# A/B of experimental builds: tools/gpu_ab_libs.sh "<name> ..." [bench args]; ab_libs/<name>.so is copied over the product library
# of the box's scratch copy of the repo before each run
set -u
O=$PWD/gpurun_out/ab; mkdir -p $O
Q="--no-cpu-baseline --no-extra-timings --verify 0 --steps 5 --warmup 2 ${2:-}"
for v in $1; do
  cp ab_libs/$v.so blingfire_amd/libblingfiretokdll.so
  timeout 300 python bench.py $Q > $O/lib_$v.json 2> $O/lib_$v.err
  python - $O/lib_$v.json $v <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print("lib", sys.argv[2], "ms/step %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernel_ms"].items()})
PY
done
