#!/bin/bash
# GPU tier, then: config 4/5 with the restructured k_prep_sp, and the default bench line with the chunked host-buffer path timed
set -u
O=$PWD/gpurun_out/r2i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
Q="--no-cpu-baseline --no-extra-timings --verify 0 --steps 5 --warmup 2"
for w in config4 config5 config3; do
  timeout 300 python bench.py $Q --workload $w > $O/$w.json 2> $O/$w.err
  python - $O/$w.json $w <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print(sys.argv[2], "docs/s %.1fM ms/step %.2f" % (j["value"] / 1e6, j["ms_per_step"]), {k: round(v, 2) for k, v in j["kernel_ms"].items()})
PY
done
timeout 600 python bench.py --no-cpu-baseline > $O/default.json 2> $O/default.err
python - $O/default.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print("default docs/s %.1fM" % (j["value"] / 1e6), j["kernel_ms"], j.get("timings"), j.get("verify"))
PY
