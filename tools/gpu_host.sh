#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/host; mkdir -p $O
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; free -g | head -2
timeout 600 python tools/host_api_probe.py ${1:-10000000} > $O/probe.txt 2>&1; grep -v amdgpu $O/probe.txt | tail -30
