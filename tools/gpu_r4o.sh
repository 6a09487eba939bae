#!/bin/bash
# round 4, session O: the split Unigram form as the default (forward kernel with sector buffers and an 8-entry record queue + k_uni_back):
# parity tests, then the sensitivity of the forward kernel to resident waves (variant bits 16..23) and to transitions per trip (bits 8..15)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4o; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 900 python -m pytest tests/test_gpu_parity_sp.py tests/test_offsets.py tests/test_gpu_api.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
Q="--no-cpu-baseline --no-extra-timings --steps 3 --warmup 1"
for spec in "config4 3" "config4 35" "config4 655363" "config4 1027" "config5 3"; do
  set -- $spec
  timeout 600 python bench.py $Q --workload $1 --variant $2 > $O/$1_v$2.json 2> $O/$1_v$2.err
  python - $O/$1_v$2.json "$1 variant $2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
