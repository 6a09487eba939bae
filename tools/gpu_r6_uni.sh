#!/bin/bash
# round 6, Unigram cut form: GPU parity tests of the _sp branch, then config 4 / 5 lines (every document verified) for the cut form, the
# round-4 kernels (variant 6) and a few emission periods / occupancies.   usage: tools/gpu_r6_uni.sh <tag> [variants...]
set -u
tag=${1:-r06_uni}; shift
O=$PWD/gpurun_out/$tag; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity_sp.py tests/test_gpu_api.py -m gpu -x -q > $O/pytest_sp.txt 2>&1; tail -3 $O/pytest_sp.txt
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); r = j["roofline"]
    print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"),
          "|", r["kernel"], "%.2f ms" % r.get("kernel_ms", 0), "| prep %.2f tok %.2f scan %.2f compact %.2f" % tuple(j["kernel_ms"][k] for k in ("prep", "tokenise", "scan", "compact")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
for w in config4 config5; do
  timeout 240 python bench.py --workload $w --no-cpu-baseline --no-extra-timings > $O/bench_$w.json 2> $O/bench_$w.err; show $O/bench_$w.json "$w cut"
done
for v in "$@"; do
  timeout 240 python bench.py --workload config4 --no-cpu-baseline --no-extra-timings --verify 0 --variant $v > $O/bench_config4_var$v.json 2> $O/bench_config4_var$v.err; show $O/bench_config4_var$v.json "config4 variant $v"
done
