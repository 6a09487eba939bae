#!/bin/bash
# GPU test tier, then one bench line per workload named ("default config2 ...")
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3q; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
for w in ${1:-default}; do
  if [ "$w" = default ]; then a=""; else a="--workload $w"; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extra-timings --steps 5 --warmup 2 $a > $O/$w.json 2> $O/$w.err
  python - $O/$w.json $w <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "verified", j.get("verified_docs"), "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernel_ms"].items()}, "status", j.get("status"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
