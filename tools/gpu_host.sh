#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/host; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "chunk or host or batch" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 600 python tools/host_api_probe.py ${1:-10000000} > $O/probe.txt 2>&1; grep -v amdgpu $O/probe.txt | tail -30
