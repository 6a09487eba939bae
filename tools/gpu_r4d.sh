#!/bin/bash
# round 4, session D: instances of the two-stage Unigram path (variants 7: 16 x 2, 8: 8 x 4, 9: 16 x 1) against the lane program (3)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4d; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 600 python -m pytest tests/test_gpu_parity_sp.py -m gpu -x -q -k "xlm or laser" > $O/pytest_sp.txt 2>&1; tail -3 $O/pytest_sp.txt
Q="--no-cpu-baseline --no-extra-timings --steps 3 --warmup 1"
for v in 7 8 9 3; do
  w=config4
  timeout 600 python bench.py $Q --workload $w --variant $v > $O/${w}_v$v.json 2> $O/${w}_v$v.err
  python - $O/${w}_v$v.json "$w variant $v" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
for v in 7 8 9; do
  cd /tmp; rm -rf /tmp/prof_d$v
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d$v/stats -o stats -- python $root/bench.py --workload config4 --variant $v --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 > /dev/null 2> $O/prof_$v.err
  cd $root
  python tools/prof_summary.py /tmp/prof_d$v $O/config4_v${v}_kernels.txt > /dev/null 2>> $O/prof_$v.err; echo "variant $v"; sed -n 3,6p $O/config4_v${v}_kernels.txt
done
timeout 300 python bench.py $Q --workload config5 --variant 7 > $O/config5_v7.json 2> $O/config5_v7.err; tail -c 300 $O/config5_v7.json; echo
BF_LEX_STATS=1 timeout 200 python tools/uni_walk_stats.py > $O/walk_stats.txt 2>&1; cat $O/walk_stats.txt
ls $O
