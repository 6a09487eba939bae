#!/bin/bash
# a quick look at k_wp_wave on the metric's corpus: time of the tokenise kernel (no verification: the parity tests are another script), scalar / vector instruction counts
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/quick_wp; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 300 python bench.py --no-cpu-baseline --no-extra-timings --verify 200000 --steps 5 --warmup 2 > $O/default.json 2> $O/default.err
python - $O/default.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print("default: value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"))
PY
cd /tmp; rm -rf /tmp/q_pmc
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/q_pmc -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 --docs 2500000 > /dev/null 2> $O/pmc.err
python - /tmp/q_pmc <<'PY'
import glob, os, sqlite3, sys
try:
    db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t.startswith("counters_collection")][0]
    for k, c, a in db.execute("select kernel_name, counter_name, avg(value) from %s where kernel_name like '%%k_wp_wave%%' group by kernel_name, counter_name" % v):
        print("2.5 M documents:", c, "%.4g" % a, "(per document %.0f)" % (a / 2.5e6))
except Exception as e: print("pmc failed", e)
PY
