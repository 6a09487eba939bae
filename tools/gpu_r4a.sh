#!/bin/bash
# round 4, session A: the merged tree (BPE wave program by default) -- GPU tier, config 3 line, issue-rate microbenchmark
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4a; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 120 tools/microbench/valu_issue 20000 > $O/valu_issue.txt 2>&1; head -12 $O/valu_issue.txt
cd /tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU --kernel-trace -d /tmp/vi_pmc -o pmc -- $root/tools/microbench/valu_issue 5000 > $O/valu_issue_pmc.log 2>&1
python - /tmp/vi_pmc $O/valu_issue_pmc.txt <<'PY'
import glob, os, sqlite3, sys
src, dst = sys.argv[1], sys.argv[2]
out = []
try:
    db = sqlite3.connect(glob.glob(os.path.join(src, "**", "*.db"), recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t.startswith("counters_collection")]
    rows = db.execute("select kernel_name, dispatch_id, counter_name, value from %s order by dispatch_id" % v[0]).fetchall()
    cur = {}
    for k, d, c, val in rows:
        cur.setdefault((d, k), {})[c] = val
    for (d, k), cs in sorted(cur.items()):
        out.append("%d %s %s" % (d, k[:40], " ".join("%s=%.0f" % (c, x) for c, x in sorted(cs.items()))))
except Exception as e:
    out.append("error: %r" % (e,))
open(dst, "w").write("\n".join(out) + "\n")
PY
cd $root
timeout 600 python bench.py --workload config3 > $O/bench_config3.json 2> $O/bench_config3.err; tail -c 600 $O/bench_config3.json; echo
timeout 600 python bench.py --workload config3 --variant 67 --no-cpu-baseline --no-extra-timings > $O/bench_config3_lane.json 2> $O/bench_config3_lane.err; tail -c 300 $O/bench_config3_lane.json; echo
ls $O
