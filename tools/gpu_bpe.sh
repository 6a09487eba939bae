#!/bin/bash
# BPE: parity of the SentencePiece-style tests, then config 3 with the arc-free k_bpe_fused (default) against the arc-writing instances (variant 5)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/bpe; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_sp.py tests/test_gpu_large_docs.py tests/test_zz_gpu_bpe_arc_pool.py tests/test_gpu_api.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
Q="--no-cpu-baseline --no-extra-timings --steps 5 --warmup 2 --workload config3"
for v in 3 5; do
  timeout 400 python bench.py $Q --variant $v > $O/config3_v$v.json 2> $O/config3_v$v.err
  python - $O/config3_v$v.json $v <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print("config3 variant", sys.argv[2], "verified", j.get("verified_docs"), "ms/step %.2f" % j["ms_per_step"], "docs/s %.1f M" % (j["value"] / 1e6), {k: round(v, 2) for k, v in j["kernel_ms"].items()}, "status", j.get("status"))
except Exception as e: print("variant", sys.argv[2], "failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
