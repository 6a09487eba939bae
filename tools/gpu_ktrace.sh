#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace) of one bench command: bash tools/gpu_ktrace.sh <name> <bench args...>
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/ktrace; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
name=$1; shift
cd /tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 3 --warmup 1 "$@" > /dev/null 2> $O/$name.err
python - /tmp/kt > $O/$name.txt 2>&1 <<'PY'
import glob, os, sqlite3, sys
db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kt[0])]
namecol = "name" if "name" in cols else "kernel_name"
rows = db.execute("select %s, count(*), avg(end-start), min(end-start), max(end-start) from %s group by %s order by avg(end-start)*count(*) desc" % (namecol, kt[0], namecol)).fetchall()
for n, c, a, mn, mx in rows[:25]:
    print("%-70s n=%-4d avg %.3f ms  min %.3f  max %.3f" % (n[:70], c, a / 1e6, mn / 1e6, mx / 1e6))
PY
cat $O/$name.txt
