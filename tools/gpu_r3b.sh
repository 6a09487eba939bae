#!/bin/bash
# round 3, session B: the pipelined wave program: parity (WordPiece tests + the 10 M-document check), A/B of units per lane / queue sizes
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3b; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity_wp.py tests/test_gpu_api.py tests/test_gpu_large_docs.py -m gpu -x -q > $O/pytest_wp.txt 2>&1; tail -3 $O/pytest_wp.txt
Q="--no-cpu-baseline --no-extra-timings --steps 5 --warmup 2"
timeout 400 python bench.py $Q > $O/default_verified.json 2> $O/default_verified.err; python - $O/default_verified.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print("default: verified", j.get("verified_docs"), "ms/step %.2f" % j["ms_per_step"], j["kernel_ms"], "status", j.get("status"))
PY
for v in 259 515 771 1027 4099 50331651; do
  timeout 300 python bench.py $Q --verify 0 --variant $v > $O/v_$v.json 2> $O/v_$v.err
  python - $O/v_$v.json $v <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print("variant", sys.argv[2], "ms/step %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernel_ms"].items()})
except Exception as e: print("variant", sys.argv[2], "failed", e)
PY
done
timeout 200 python tools/wave_stats.py 300000 > $O/stats.txt 2>&1; tail -2 $O/stats.txt
