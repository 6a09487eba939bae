#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, (optional) rocprof.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== $(date) host: $(nproc) cores"; grep -m1 "model name" /proc/cpuinfo
  rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
} > gpurun_out/box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
for extra in "$@"; do
  eval "$extra"
done
