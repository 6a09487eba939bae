#!/bin/bash
# a quick look at k_bpe_wave on config 3: kernel times, instruction counts, the program's own counters
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/quick_bpe; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 300 python bench.py --workload config3 --no-cpu-baseline --no-extra-timings --verify 100000 --steps 5 --warmup 2 > $O/config3.json 2> $O/config3.err
python - $O/config3.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print("config3: value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"))
PY
cd /tmp; rm -rf /tmp/q_pmc
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY --kernel-trace -d /tmp/q_pmc -o pmc -- python $root/bench.py --workload config3 --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 > /dev/null 2> $O/pmc.err
python - /tmp/q_pmc <<'PY'
import glob, os, sqlite3, sys
try:
    db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t.startswith("counters_collection")][0]
    for k, c, a in db.execute("select kernel_name, counter_name, avg(value) from %s where kernel_name like '%%k_bpe_wave%%' group by kernel_name, counter_name" % v):
        print("1 M documents:", c, "%.4g" % a, "(per document %.0f)" % (a / 1e6))
except Exception as e: print("pmc failed", e)
PY
cd $root
BF_LEX_STATS=1 timeout 120 python tools/bpe_wave_stats.py 200000 2>&1 | grep -v amdgpu
