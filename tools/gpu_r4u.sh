#!/bin/bash
# round 4, session U: the _sp prologue (k_prep_sp8) compiled for 6 (the compiler's choice, 73 VGPRs) / 7 (72) / 8 (64 VGPRs, 8 bytes of scratch) waves per SIMD:
# BfSetVariant bits 24..27; configs 4 and 5, 10 M documents verified
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4u; mkdir -p $O
Q="--no-cpu-baseline --no-extra-timings --steps 3 --warmup 1"
for spec in "config4 3" "config4 117440515" "config4 134217731" "config5 3" "config5 117440515"; do
  set -- $spec
  timeout 600 python bench.py $Q --workload $1 --variant $2 > $O/$1_v$2.json 2> $O/$1_v$2.err
  python - $O/$1_v$2.json "$1 variant $2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), {k: round(v, 2) for k, v in j.get("kernel_ms", {}).items()})
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
