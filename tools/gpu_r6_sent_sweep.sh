#!/bin/bash
set -u
tag=${1:-r06_sent}; O=$PWD/gpurun_out/$tag; mkdir -p $O
for v in 0x40000000 0x3000 0x5000 0x7000 0; do timeout 300 python tools/bench_sentences.py 262144 $v 2>&1 | grep -v "amdgpu.ids\|Warn\|dt, do"; done | tee $O/sent_sweep.txt
