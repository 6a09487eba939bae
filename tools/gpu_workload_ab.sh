#!/bin/bash
# A/B of one workload over kernel variants.   usage: tools/gpu_workload_ab.sh <workload> "<variants>" [pytest file]
set -u
export TMPDIR=/tmp
w=${1:-config3}
O=$PWD/gpurun_out/workload_ab; mkdir -p $O
Q="--no-cpu-baseline --no-extra-timings --steps 5 --warmup 2 --workload $w"
show() { python - $1 "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "verified", j.get("verified_docs"), "ms/step %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernel_ms"].items()}, "status", j.get("status"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
if [ -n "${3:-}" ]; then timeout 900 python -m pytest $3 -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt; fi
timeout 400 python bench.py $Q > $O/${w}_default.json 2> $O/${w}_default.err; show $O/${w}_default.json "$w default"
for v in ${2:-}; do
  timeout 300 python bench.py $Q --variant $v > $O/${w}_v_$v.json 2> $O/${w}_v_$v.err; show $O/${w}_v_$v.json "$w variant $v"
done
