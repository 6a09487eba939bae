#!/bin/bash
# round 3, session A: first run of the wave kernel (bf_wave.h) on the GPU: WordPiece parity tests, A/B against the lane-per-document
# kernels (variant 2) and between queue / ring configurations, rocprof summary of the default line.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3a; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 500 python -m pytest tests/test_gpu_parity_wp.py tests/test_gpu_api.py tests/test_gpu_large_docs.py -m gpu -x -q > $O/pytest_wp.txt 2>&1; tail -3 $O/pytest_wp.txt
timeout 400 python bench.py --no-cpu-baseline --no-extra-timings --steps 5 --warmup 2 > $O/default_verified.json 2> $O/default_verified.err; tail -c 600 $O/default_verified.json; tail -3 $O/default_verified.err
Q="--no-cpu-baseline --no-extra-timings --verify 0 --steps 5 --warmup 2"
for v in 2 3 259 515 771 1027 4099 16387 33554435 50331651; do
  timeout 300 python bench.py $Q --variant $v > $O/v_$v.json 2> $O/v_$v.err
  python - $O/v_$v.json $v <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print("variant", sys.argv[2], "ms/step %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernel_ms"].items()})
except Exception as e: print("variant", sys.argv[2], "failed", e)
PY
done
timeout 300 python bench.py $Q --workload config2 > $O/config2.json 2> $O/config2.err; tail -c 400 $O/config2.json
timeout 200 python tools/wave_stats.py 300000 > $O/stats.txt 2>&1; tail -2 $O/stats.txt; timeout 200 python tools/wave_stats.py 300000 259 >> $O/stats.txt 2>&1; tail -1 $O/stats.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_a/stats -o stats -- python $root/bench.py $Q > $O/traced.json 2> $O/traced.err
cd $root
python tools/prof_summary.py /tmp/prof_a $O/default_prof.txt > /dev/null 2> $O/prof_summary.err
ls -la $O
