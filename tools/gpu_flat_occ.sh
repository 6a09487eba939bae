#!/bin/bash
# the flat program compiled for 7 / 6 / 5 waves per SIMD (a build with BF_EXPERIMENTS) against the wave program, 2.5 M documents of the metric's corpus
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/flat_occ; mkdir -p $O
run() { # name variant [extra args]
  timeout 300 python bench.py --no-cpu-baseline --no-extra-timings --verify ${VERIFY:-200000} --steps 4 --warmup 2 --docs ${3:-2500000} --variant $2 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json $1 <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), {k: round(v, 3) for k, v in j.get("kernel_ms").items()})
except Exception as e: print(sys.argv[2], "failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
}
for v in "$@"; do run v$v $v; done
