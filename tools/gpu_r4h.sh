#!/bin/bash
# round 4, session H: k_compact_text with the spans in registers
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4h; mkdir -p $O
timeout 600 python -m pytest tests/test_offsets.py tests/test_gpu_api.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
Q="--no-cpu-baseline --no-extra-timings --steps 5 --warmup 2"
timeout 400 python bench.py $Q --offsets > $O/default_offsets.json 2> $O/default_offsets.err
timeout 400 python bench.py $Q --offsets --workload config2 > $O/config2_offsets.json 2> $O/config2_offsets.err
for f in default_offsets config2_offsets; do python - $O/$f.json $f <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), j["verify"].get("offsets"), "status", j.get("status"), j.get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
