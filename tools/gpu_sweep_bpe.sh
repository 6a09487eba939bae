#!/bin/bash
# config 3 with the variants given as arguments (kernel time only; a 100 k-document check each)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/sweep_bpe; mkdir -p $O
for v in "$@"; do
  timeout 300 python bench.py --workload config3 --no-cpu-baseline --no-extra-timings --verify 100000 --steps 5 --warmup 2 --variant $v > $O/v$v.json 2> $O/v$v.err
  python - $O/v$v.json $v <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print("variant", sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), "tokenise %.2f" % j["kernel_ms"]["tokenise"])
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
