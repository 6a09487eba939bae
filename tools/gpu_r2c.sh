#!/bin/bash
# round 2, GPU session C: two-level lexer A/B + event-threshold sweep, Unigram occupancy sweep
set -u
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "wp or api or offsets or words or wrapper" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
B="python bench.py --docs 1250000 --no-cpu-baseline --no-extra-timings --verify 20000"
# lexer: 7 = general machine; 3 = two-level (event threshold 16); thresholds 8 / 24 / 32 / 4; unroll 2 / 4 with threshold 16
for v in 7 3 $((3 + 8*256)) $((3 + 24*256)) $((3 + 32*256)) $((3 + 4*256)) $((3 + 16*256 + 2*1073741824)) $((3 + 16*256 + 3*1073741824)); do
  timeout 200 $B --variant $v > $O/lex_v$v.json 2>> $O/err.txt
done
# the same on the full corpus for the two best guesses
timeout 400 python bench.py --no-cpu-baseline --no-extra-timings --verify 100000 > $O/lex_full_default.json 2>> $O/err.txt
# Unigram: resident waves per CU 20 (default) / 16 / 12 / 8 / 4
U="python bench.py --workload config4 --docs 1250000 --no-cpu-baseline --no-extra-timings --verify 20000"
for w in 0 16 12 8 4; do timeout 200 $U --variant $((3 + w*65536)) > $O/uni_w$w.json 2>> $O/err.txt; done
python - <<'PY' > $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r2c/*.json")):
    try:
        r = json.load(open(f))
        print("%-28s %7.1f M docs/s  tok %.3f ms  prep %.3f  total %.3f  verified %d" % (f.split("/")[-1], r["value"] / 1e6, r["kernel_ms"]["tokenise"], r["kernel_ms"]["prep"], r["kernel_ms"]["total"], r["verified_docs"]))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt
