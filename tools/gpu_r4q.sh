#!/bin/bash
# round 4, session Q: k_wp_wave (shipped instance) at 8 .. 2 workgroups per CU (= waves per SIMD; variant bits 24..29): how the time follows the
# number of resident waves says whether the kernel waits (time ~ 1 / waves) or issues (flat)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4q; mkdir -p $O
Q="--no-cpu-baseline --no-extra-timings --steps 6 --warmup 2 --verify 0 --docs 2500000"
for w in 8 7 6 5 4 3 2; do
  v=$((w << 24))
  timeout 300 python bench.py $Q --variant $v > $O/wp_w$w.json 2> $O/wp_w$w.err
  python - $O/wp_w$w.json "2.5 M docs, $w workgroups per CU" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "tokenise %.3f ms" % j["kernel_ms"]["tokenise"], "status", j.get("status"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
