#!/bin/bash
# round 6, the words modes: the long-document threshold (BfSetVariant bits 12..15: 8 << k characters) on config 1's lines and on 1 M synthetic lines
set -u
tag=${1:-r06_words_sweep}; O=$PWD/gpurun_out/$tag; mkdir -p $O
for v in 0 0x40000000 0x1000 0x2000 0x3000 0x4000 0x5000; do
  timeout 120 python tools/bench_words.py 10000 $v config1 2>&1 | tail -1
done | tee $O/sweep_config1.txt
for v in 0 0x40000000 0x3000 0x4000; do
  timeout 200 python tools/bench_words.py 1000000 $v 2>&1 | tail -1
done | tee $O/sweep_1m.txt
for n in 100 1000 100000; do
  for v in 0 0x40000000; do timeout 200 python tools/bench_words.py $n $v 2>&1 | tail -1; done
done | tee $O/sweep_sizes.txt
