#!/bin/bash
# A/B of two builds of the library on ONE box: usage tools/gpu_ab.sh <tag> <a.so> <b.so> [bench args]; three alternating rounds
set -u
tag=$1; A=$2; B=$3; shift 3
O=$PWD/gpurun_out/$tag; mkdir -p $O
cp blingfire_amd/libblingfiretokdll.so /tmp/keep.so
for round in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then cp $A blingfire_amd/libblingfiretokdll.so; else cp $B blingfire_amd/libblingfiretokdll.so; fi
    timeout 300 python bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 20 --warmup 5 "$@" > $O/ab_${v}_$round.json 2> $O/ab_${v}_$round.err
    python - $O/ab_${v}_$round.json "$v round $round" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); r = j["roofline"]
    print(sys.argv[2], "ms/step %.2f" % j["ms_per_step"], "|", r["kernel"], "%.2f ms" % r.get("kernel_ms", 0), "| prep %.2f tok %.2f scan %.2f compact %.2f" % tuple(j["kernel_ms"][k] for k in ("prep", "tokenise", "scan", "compact")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
  done
done
cp /tmp/keep.so blingfire_amd/libblingfiretokdll.so
