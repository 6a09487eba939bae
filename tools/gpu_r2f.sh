#!/bin/bash
# round 2, GPU session F: fast-forward once per vote + PLAIN instance; A/B on the 1.25 M-doc shard, then the default line
set -u
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "wp or api or offsets or words or wrapper or large" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
B="python bench.py --docs 1250000 --no-cpu-baseline --no-extra-timings --verify 20000"
for v in 3 4 8 $((3 + 3*1048576)) $((3 + 5*1048576)) $((3 + 8*256)) $((3 + 32*256)); do timeout 200 $B --variant $v > $O/lex_v$v.json 2>> $O/err.txt; done
timeout 900 python bench.py --no-cpu-baseline --no-extra-timings > $O/bench_default.json 2>> $O/err.txt
timeout 100 python bench.py --workload config1 > $O/bench_config1.json 2>> $O/err.txt
timeout 100 python bench.py --workload config1 --docs 1000000 --verify 20000 > $O/bench_config1_1M.json 2>> $O/err.txt
python - <<'PY' > $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r2f/*.json")):
    try:
        r = json.load(open(f))
        km = r.get("kernel_ms", {})
        print("%-28s %9.1f M/s  tok %.3f ms  prep %.3f  total %.3f  verified %d  %s" % (f.split("/")[-1], r["value"] / 1e6, km.get("tokenise", 0), km.get("prep", 0), km.get("total", r.get("gpu_ms_per_step", 0)), r["verified_docs"], r.get("cpu_baseline", {}).get("value")))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt; tail -3 $O/pytest_gpu.log; tail -5 $O/err.txt | grep -v amdgpu.ids
