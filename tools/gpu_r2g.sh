#!/bin/bash
# round 2, GPU session G: fast-forward back inside the step; prep_sp ASCII-window path; unroll 4 vs 5
set -u
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
B="python bench.py --docs 1250000 --no-cpu-baseline --no-extra-timings --verify 20000"
for v in 3 4 $((3 + 5*1048576)) $((3 + 6*1048576)); do timeout 200 $B --variant $v > $O/lex_v$v.json 2>> $O/err.txt; done
timeout 900 python bench.py --no-cpu-baseline --no-extra-timings --verify 100000 > $O/bench_default.json 2>> $O/err.txt
timeout 900 python bench.py --no-cpu-baseline --no-extra-timings --verify 100000 --variant $((3 + 5*1048576)) > $O/bench_default_u5.json 2>> $O/err.txt
for w in config4 config5; do timeout 600 python bench.py --workload $w --docs 2500000 --no-cpu-baseline --no-extra-timings > $O/bench_${w}_2500k.json 2>> $O/err.txt; done
timeout 300 python bench.py --workload config3 --no-cpu-baseline --no-extra-timings > $O/bench_config3.json 2>> $O/err.txt
python - <<'PY' > $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r2g/*.json")):
    try:
        r = json.load(open(f))
        km = r.get("kernel_ms", {})
        print("%-28s %9.1f M/s  tok %.3f ms  prep %.3f  total %.3f  verified %d" % (f.split("/")[-1], r["value"] / 1e6, km.get("tokenise", 0), km.get("prep", 0), km.get("total", 0), r["verified_docs"]))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt; tail -3 $O/pytest_gpu.log
