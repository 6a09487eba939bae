#!/bin/bash
# round 4, session S: where k_wp_wave waits.  The instances without vocabulary walks (configuration 14) and without walks and retire pass (15) -- wrong results by
# design -- at 8 / 4 / 2 workgroups per CU next to the whole kernel (12 = TRIM 0, the instance the phase instances are built from): T = a + b / waves per phase by difference
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4s; mkdir -p $O
Q="--no-cpu-baseline --no-extra-timings --steps 5 --warmup 2 --verify 0 --docs 2500000"
for cfg in 12 14 15; do for w in 8 4 2; do
  v=$(( (w << 24) | (cfg << 8) ))
  timeout 300 python bench.py $Q --variant $v > $O/wp_c${cfg}_w$w.json 2> $O/wp_c${cfg}_w$w.err
  python - $O/wp_c${cfg}_w$w.json "configuration $cfg, $w workgroups per CU" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "tokenise %.3f ms" % j["kernel_ms"]["tokenise"])
except Exception as e: print(sys.argv[2], "failed", e)
PY
done; done
