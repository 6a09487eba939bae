#!/bin/bash
# round 4, session R: k_wp_wave, the retire pass with one wait (TRIM 8: configuration 13) against the shipped instance (TRIM 7), and what the stores to
# the provisional homes cost (configuration 3: none of them, wrong results by design, not verified)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4r; mkdir -p $O
Q="--no-cpu-baseline --no-extra-timings --steps 6 --warmup 2 --docs 2500000"
for spec in "0 200000" "3328 200000" "768 0" "0 0" "3328 0"; do
  set -- $spec
  timeout 300 python bench.py $Q --variant $1 --verify $2 > $O/wp_v$1_$2.json 2> $O/wp_v$1_$2.err
  python - $O/wp_v$1_$2.json "2.5 M docs, variant $1 verify $2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "tokenise %.3f ms" % j["kernel_ms"]["tokenise"], "verified", j.get("verified_docs"), "status", j.get("status"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
