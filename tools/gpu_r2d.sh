#!/bin/bash
# round 2, GPU session D: full GPU tests, lexer transitions-per-vote sweep (two-level form), packed-record Unigram kernel
set -u
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
B="python bench.py --docs 1250000 --no-cpu-baseline --no-extra-timings --verify 20000"
for u in 0 3 5 6 8; do timeout 200 $B --variant $((3 + u*1048576)) > $O/lex_u$u.json 2>> $O/err.txt; done
U="python bench.py --workload config4 --docs 1250000 --no-cpu-baseline --no-extra-timings --verify 20000"
for v in 3 $((3 + 2*256)) $((3 + 4*256)) $((3 + 12*65536)) $((3 + 8*65536)); do timeout 200 $U --variant $v > $O/uni_v$v.json 2>> $O/err.txt; done
timeout 200 python bench.py --workload config5 --docs 1250000 --no-cpu-baseline --no-extra-timings --verify 20000 > $O/uni_c5.json 2>> $O/err.txt
python - <<'PY' > $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r2d/*.json")):
    try:
        r = json.load(open(f))
        print("%-28s %7.1f M docs/s  tok %.3f ms  prep %.3f  total %.3f  verified %d" % (f.split("/")[-1], r["value"] / 1e6, r["kernel_ms"]["tokenise"], r["kernel_ms"]["prep"], r["kernel_ms"]["total"], r["verified_docs"]))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt; tail -3 $O/pytest_gpu.log; tail -5 $O/err.txt
