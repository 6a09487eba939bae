#!/bin/bash
# parity for given BF_LEX_VARIANT values then a bench sweep.  usage: gpu_sweep.sh "<parity variants>" "<bench variants>" [bench args]
pv=$1; bv=$2; shift; shift
mkdir -p gpurun_out
for v in $pv; do
  echo "== parity variant $v"; BF_LEX_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_parity_wp.py -x -q 2>&1 | grep -E "passed|failed|AssertionError" | head -3
done
bash tools/gpu_variants.sh "$bv" "$@"
