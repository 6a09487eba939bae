#!/bin/bash
# instruction counts of the config-4 kernels (2.5 M documents per launch)
set -u
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; rm -rf /tmp/q_pmc
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace -d /tmp/q_pmc -o pmc -- python $root/bench.py --workload config4 --no-cpu-baseline --no-extra-timings --verify 0 --steps 1 --warmup 1 --docs 2500000 > /dev/null 2> /tmp/pmc.err
python - /tmp/q_pmc <<'PY'
import glob, os, sqlite3, sys
try:
    db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t.startswith("counters_collection")][0]
    for k, c, a in db.execute("select kernel_name, counter_name, avg(value) from %s where kernel_name like '%%k_prep_sp%%' or kernel_name like '%%unigram%%' group by kernel_name, counter_name" % v):
        print(k[:40], c, "%.4g" % a, "(per document %.0f)" % (a / 2.5e6))
except Exception as e: print("pmc failed", e)
PY
