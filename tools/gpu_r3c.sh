#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3c; mkdir -p $O
for v in -1 259 1027; do timeout 200 python tools/wave_stats.py 1000000 $v > $O/stats_$v.txt 2>&1; tail -3 $O/stats_$v.txt; done
