#!/bin/bash
# round 4, session N: split Unigram (variant bit 0x20) with the backward kernel reading 64-byte record sectors and the forward kernel's
# record queue at 4 / 8 / 16 entries (variant bits 8..15 = 4 / 8 / 0)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4n; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
Q="--no-cpu-baseline --no-extra-timings --steps 3 --warmup 1"
for spec in "config4 3" "config4 35" "config4 1059" "config4 2083" "config5 35" "config5 2083"; do
  set -- $spec
  timeout 600 python bench.py $Q --workload $1 --variant $2 > $O/$1_v$2.json 2> $O/$1_v$2.err
  python - $O/$1_v$2.json "$1 variant $2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
cd /tmp
for v in 35 1059 2083; do
rm -rf /tmp/q_st; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/q_st/stats -o stats -- python $root/bench.py --workload config4 --variant $v --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 --docs 10000000 > /dev/null 2> /tmp/st.err
python $root/tools/prof_summary.py /tmp/q_st $O/config4_v${v}_kernels.txt > /dev/null 2> $O/summary.err; head -5 $O/config4_v${v}_kernels.txt | tail -3 | cut -c1-110
done
