#!/bin/bash
# round 4, session P: (1) GPU tests of the ignore-case / moore-multi-dfa models, (2) the TRIM bits of k_wp_wave (bf_wave_body.h) against the shipped
# instance on the metric's corpus: variant bits 8..11 = 0 shipped, 7 TRIM 1 (no settle at the top of a trip), 9 TRIM 3 (+ token writes by selects),
# 10 TRIM 4 (fill as nested loops), 11 TRIM 7 (all).  Every line is verified against the compiled reference (200,000 documents).
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4p; mkdir -p $O
timeout 600 python -m pytest tests/test_ignore_case.py -m gpu -x -q > $O/pytest_ignore_case.txt 2>&1; tail -3 $O/pytest_ignore_case.txt
Q="--no-cpu-baseline --no-extra-timings --steps 6 --warmup 2 --verify 200000"
for v in 0 1792 2304 2560 2816 0 2816; do
  timeout 300 python bench.py $Q --docs 2500000 --variant $v > $O/wp_v$v.json 2> $O/wp_v$v.err
  python - $O/wp_v$v.json "2.5 M docs, variant $v" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.3f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
for v in 0 2816; do
  timeout 300 python bench.py $Q --variant $v > $O/wp10m_v$v.json 2> $O/wp10m_v$v.err
  python - $O/wp10m_v$v.json "10 M docs, variant $v" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.3f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
