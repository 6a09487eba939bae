#!/bin/bash
# round 6, BPE: GPU parity tests of the _sp branch, then config 3 with and without the word table (every document verified)
set -u
tag=${1:-r06_bpe}; shift
O=$PWD/gpurun_out/$tag; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity_sp.py tests/test_zz_gpu_bpe_arc_pool.py -m gpu -x -q -k "bpe or adversarial or corpus" > $O/pytest_bpe.txt 2>&1; tail -3 $O/pytest_bpe.txt
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); r = j["roofline"]
    print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"),
          "|", r["kernel"], "%.2f ms" % r.get("kernel_ms", 0), "| prep %.2f tok %.2f scan %.2f compact %.2f" % tuple(j["kernel_ms"][k] for k in ("prep", "tokenise", "scan", "compact")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
timeout 300 python bench.py --workload config3 --no-cpu-baseline --no-extra-timings > $O/bench_config3.json 2> $O/bench_config3.err; show $O/bench_config3.json "config3"
for v in 1048579 "$@"; do
  timeout 300 python bench.py --workload config3 --no-cpu-baseline --no-extra-timings --verify 0 --variant $v > $O/bench_config3_var$v.json 2> $O/bench_config3_var$v.err; show $O/bench_config3_var$v.json "config3 variant $v"
done
timeout 300 python bench.py --workload config3 --model roberta.bin --no-cpu-baseline --no-extra-timings > $O/bench_config3_roberta.json 2> $O/bench_config3_roberta.err; show $O/bench_config3_roberta.json "config3 corpus, roberta.bin"
