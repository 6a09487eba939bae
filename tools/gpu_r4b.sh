#!/bin/bash
# round 4, session B: the one-wave-per-document BPE program on the device (long runs, pool growth), roberta through the wave program
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_sp.py tests/test_zz_gpu_bpe_arc_pool.py tests/test_offsets.py -m gpu -x -q --durations=12 > $O/pytest_bpe.txt 2>&1; tail -25 $O/pytest_bpe.txt
timeout 300 python tools/bpe_long_runs.py > $O/long_runs.txt 2>&1; cat $O/long_runs.txt
timeout 600 python bench.py --workload config3 --no-cpu-baseline --no-extra-timings > $O/bench_config3.json 2> $O/bench_config3.err; tail -c 400 $O/bench_config3.json; echo
