#!/bin/bash
# round 4, session E: the Unigram lane program with the hot tables in LDS (default) against the plain one (variant 6)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4e; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 600 python -m pytest tests/test_gpu_parity_sp.py tests/test_offsets.py tests/test_multi_device.py -m gpu -x -q > $O/pytest_sp.txt 2>&1; tail -3 $O/pytest_sp.txt
Q="--no-cpu-baseline --no-extra-timings --steps 3 --warmup 1"
for w in config4 config5; do
  for v in 3 6; do
    timeout 600 python bench.py $Q --workload $w --variant $v > $O/${w}_v$v.json 2> $O/${w}_v$v.err
    python - $O/${w}_v$v.json "$w variant $v" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
  done
done
ls $O
