#!/bin/bash
# A/B of k_wp_wave configurations: WordPiece GPU parity tests, the default workload (2 M documents verified), variants, config2,
# the counters of the STATS instance.   usage: tools/gpu_wave_ab.sh "<variants>"
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/wave_ab; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_wp.py tests/test_gpu_large_docs.py -m gpu -x -q > $O/pytest_wp.txt 2>&1; tail -3 $O/pytest_wp.txt
Q="--no-cpu-baseline --no-extra-timings --steps 5 --warmup 2"
show() { python - $1 "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "verified", j.get("verified_docs"), "ms/step %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernel_ms"].items()}, "status", j.get("status"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
timeout 400 python bench.py $Q --verify 2000000 > $O/default.json 2> $O/default.err; show $O/default.json default
for v in ${1:-}; do
  timeout 300 python bench.py $Q --verify 0 --variant $v > $O/v_$v.json 2> $O/v_$v.err; show $O/v_$v.json "variant $v"
done
timeout 300 python bench.py $Q --workload config2 > $O/config2.json 2> $O/config2.err; show $O/config2.json config2
timeout 200 python tools/wave_stats.py 2000000 2>&1 | tail -1
