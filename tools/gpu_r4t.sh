#!/bin/bash
# round 4, session T: the N > 1 form of the bench on the 1-GPU box (two ranks sharing the device, gloo: BF_BENCH_SHARE_GPU=1, testing only) with the driver's command line
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4t; mkdir -p $O
BF_BENCH_SHARE_GPU=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2ranks.json 2> $O/bench_2ranks.err
tail -c 1500 $O/bench_2ranks.json; echo; tail -5 $O/bench_2ranks.err
BF_BENCH_SHARE_GPU=1 timeout 300 python bench.py --inproc --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-extra-timings > $O/bench_inproc2.json 2> $O/bench_inproc2.err
tail -c 600 $O/bench_inproc2.json; echo; tail -3 $O/bench_inproc2.err
