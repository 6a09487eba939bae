#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/single; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity_wp.py tests/test_golden_api.py tests/test_reference_wrapper.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/bench_single_calls.py > $O/single_calls.txt 2>&1; cat $O/single_calls.txt | grep -v amdgpu
