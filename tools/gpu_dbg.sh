cd $GRAFT_REPO_ROOT
python bench.py --docs 20000 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-timings > gpurun_out/dbg1.json 2> gpurun_out/dbg1.err; echo rc $?; tail -3 gpurun_out/dbg1.err; tail -c 300 gpurun_out/dbg1.json
BF_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --docs 40000 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-timings > gpurun_out/dbg2.json 2> gpurun_out/dbg2.err; echo rc $?; grep -v "^W09\|^\[W\|amdgpu.ids" gpurun_out/dbg2.err | head -40
BF_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --inproc --docs 40000 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-timings > gpurun_out/dbg3.json 2> gpurun_out/dbg3.err; echo rc $?; tail -5 gpurun_out/dbg3.err; tail -c 300 gpurun_out/dbg3.json
