#!/bin/bash
# extra PMC passes for one kernel: tools/gpu_pmc.sh <tag> "<bench args>" "<counters pass 1>" "<counters pass 2>" ...
set -u
tag=$1; args=$2; shift; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp
i=0
for c in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c -d "$out/p$i" -o pmc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --verify 0 $args > "$out/p$i.log" 2>&1
done
cd "$GRAFT_REPO_ROOT"
python - "$out" <<'PY'
import glob, sqlite3, sys
for db in sorted(glob.glob(sys.argv[1] + "/p*/*.db")):
    c = sqlite3.connect(db)
    for k, n, v, d in c.execute("select kernel_name, counter_name, avg(value), avg(duration) from counters_collection group by kernel_name, counter_name order by avg(duration) desc"):
        if "lex" in k or "seg" in k or "bpe" in k or "prep" in k:
            print("%-40s %-34s %16.0f  avg %.3f ms" % (k.split("(")[0][-40:], n, v, d / 1e6))
PY
find "$out" -size +8M -delete
