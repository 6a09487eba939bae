#!/bin/bash
# round 6, the metric's path: WordPiece GPU tests, then the default line (every document verified) and its offsets form
set -u
tag=${1:-r06_wp}; O=$PWD/gpurun_out/$tag; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_wp.py tests/test_offsets.py -m gpu -x -q > $O/pytest_wp.txt 2>&1; tail -3 $O/pytest_wp.txt
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); r = j["roofline"]
    print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"),
          "|", r["kernel"], "%.2f ms" % r.get("kernel_ms", 0), "| prep %.2f tok %.2f scan %.2f compact %.2f" % tuple(j["kernel_ms"][k] for k in ("prep", "tokenise", "scan", "compact")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
timeout 400 python bench.py --no-cpu-baseline --no-extra-timings > $O/bench_default.json 2> $O/bench_default.err; show $O/bench_default.json "default"
timeout 400 python bench.py --no-cpu-baseline --no-extra-timings --steps 20 --warmup 5 --verify 0 > $O/bench_default2.json 2> $O/bench_default2.err; show $O/bench_default2.json "default (20 steps)"
timeout 400 python bench.py --no-cpu-baseline --no-extra-timings --offsets --verify 2000000 > $O/bench_offsets.json 2> $O/bench_offsets.err; show $O/bench_offsets.json "offsets"
timeout 300 python bench.py --no-cpu-baseline --no-extra-timings --workload config2 > $O/bench_config2.json 2> $O/bench_config2.err; show $O/bench_config2.json "config2"
