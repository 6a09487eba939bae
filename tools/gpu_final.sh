#!/bin/bash
# round-end measurement set: default bench (with the CPU reference leg), the other configs, the N>1 code path on one GPU
mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/final_headline.json 2> gpurun_out/final_headline.err
for w in config2 config3 config4 config5; do
  timeout 400 python bench.py --workload $w > gpurun_out/final_$w.json 2> gpurun_out/final_$w.err
done
BF_BENCH_SHARE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --docs-per-gpu 300000 > gpurun_out/final_n2_shared.json 2> gpurun_out/final_n2_shared.err
for f in gpurun_out/final_*.json; do echo "== $f"; tail -1 $f | cut -c1-400; done
