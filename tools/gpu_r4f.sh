#!/bin/bash
# round 4, session F: offsets through the wave program (OFFS instance), sharded host path, Unigram with the score array, GPU tier
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4f; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
Q="--no-cpu-baseline --no-extra-timings --steps 5 --warmup 2"
timeout 400 python bench.py $Q > $O/default_ids.json 2> $O/default_ids.err
timeout 400 python bench.py $Q --offsets > $O/default_offsets.json 2> $O/default_offsets.err
timeout 400 python bench.py $Q --offsets --variant 2 > $O/default_offsets_lane.json 2> $O/default_offsets_lane.err
for f in default_ids default_offsets default_offsets_lane; do python - $O/$f.json $f <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), j["verify"].get("offsets"), "status", j.get("status"), j.get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
for w in config4 config5; do
  timeout 600 python bench.py $Q --steps 3 --warmup 1 --workload $w > $O/$w.json 2> $O/$w.err
  python - $O/$w.json $w <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
bash tools/gpu_host.sh > $O/host.txt 2>&1; tail -20 $O/host.txt
ls $O
