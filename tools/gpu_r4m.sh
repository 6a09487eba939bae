#!/bin/bash
# round 4, session M: the Unigram backward pass as its own kernel (variant bit 0x20: k_seg_unigram_lane<3, true> + k_uni_back)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4m; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
Q="--no-cpu-baseline --no-extra-timings --steps 3 --warmup 1"
for spec in "config4 3" "config4 35" "config5 3" "config5 35"; do
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
for v in 3 35; do
  rm -rf /tmp/q_pmc$v
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d /tmp/q_pmc$v -o pmc -- python $root/bench.py --workload config4 --variant $v --no-cpu-baseline --no-extra-timings --verify 0 --steps 1 --warmup 1 --docs 2500000 > /dev/null 2> /tmp/pmc$v.err
  python - /tmp/q_pmc$v $v <<'PY'
import glob, os, sqlite3, sys
try:
    db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t.startswith("counters_collection")][0]
    for k, c, a in db.execute("select kernel_name, counter_name, avg(value) from %s where kernel_name like '%%uni%%' group by kernel_name, counter_name" % v):
        print("variant", sys.argv[2], k[:50], c, "%.4g" % a)
except Exception as e: print("pmc failed", e); print(open("/tmp/pmc%s.err" % sys.argv[2]).read()[-800:])
PY
done
rm -rf /tmp/q_st; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/q_st/stats -o stats -- python $root/bench.py --workload config4 --variant 35 --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 --docs 10000000 > /dev/null 2> /tmp/st.err
python $root/tools/prof_summary.py /tmp/q_st $O/config4_v35_kernels.txt > /dev/null 2> $O/summary.err; head -8 $O/config4_v35_kernels.txt | cut -c1-110
