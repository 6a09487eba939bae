#!/bin/bash
# the GPU test tier + smoke, as the driver runs them
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/tests; mkdir -p $O
timeout ${1:-1500} python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
