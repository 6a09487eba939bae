#!/bin/bash
# round 4, session C: Unigram in two stages (k_uni_walk + k_uni_dp) on the device: parity, configs 4 / 5 against the lane program (variant 6), kernel split
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4c; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 900 python -m pytest tests/test_gpu_parity_sp.py tests/test_offsets.py tests/test_gpu_api.py -m gpu -x -q --durations=8 > $O/pytest_sp.txt 2>&1; tail -14 $O/pytest_sp.txt
Q="--no-cpu-baseline --no-extra-timings --steps 4 --warmup 1"
for w in config4 config5; do
  for v in 3 6; do
    timeout 600 python bench.py $Q --workload $w --variant $v > $O/${w}_v$v.json 2> $O/${w}_v$v.err
    python - $O/${w}_v$v.json "$w variant $v" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "kernel", j["roofline"]["kernel"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
  done
done
cd /tmp
rm -rf /tmp/prof_c4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4/stats -o stats -- python $root/bench.py --workload config4 --no-cpu-baseline --no-extra-timings --verify 0 --steps 3 --warmup 1 > /dev/null 2> $O/prof_c4.err
cd $root
python tools/prof_summary.py /tmp/prof_c4 $O/config4_kernels.txt > /dev/null 2>> $O/prof_c4.err; head -30 $O/config4_kernels.txt
ls $O
