#!/bin/bash
# counters of the flat program (variant $1) on 2.5 M documents of the metric's corpus
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/flat_pmc; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
V=${1:-3}
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  rm -rf /tmp/fl_pmc
  rm -f $O/pmc_v$V.txt.tmp
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/fl_pmc -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 --docs 2500000 --variant $V > /dev/null 2>> $O/pmc.err
  python - /tmp/fl_pmc >> $O/pmc_v$V.txt 2>&1 <<'PY'
import glob, os, sqlite3, sys
try:
    db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t.startswith("counters_collection")][0]
    for k, c, a in db.execute("select kernel_name, counter_name, avg(value) from %s where kernel_name like '%%k_wp_%%' group by kernel_name, counter_name" % v):
        if "flat" in k or "merge" in k or "units" in k: print(k[:36], c, "%.4g" % a, "(per document %.1f)" % (a / 2.5e6))
except Exception as e: print("pmc failed", e)
PY
done
cat $O/pmc_v$V.txt
