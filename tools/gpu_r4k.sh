#!/bin/bash
# round 4, session K: k_prep_sp8 with the next document prefetched, at the compiler's register count and capped to 8 waves per SIMD (variant bit 0x100)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_sp.py tests/test_offsets.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
Q="--no-cpu-baseline --no-extra-timings --steps 3 --warmup 1"
for spec in "config4 3" "config4 259" "config5 3" "config5 259" "config3 3" "config3 259"; do
  set -- $spec
  timeout 600 python bench.py $Q --workload $1 --variant $2 > $O/$1_v$2.json 2> $O/$1_v$2.err
  python - $O/$1_v$2.json "$1 variant $2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
