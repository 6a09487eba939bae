#!/bin/bash
# round 2, GPU session E: bert_base_tok.bin is here -- full GPU tests, the default bench line, configs 3-5, PMC of the packed-record Unigram kernel
set -u
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload config2 > $O/bench_config2.json 2>> $O/err.txt
timeout 300 python bench.py --workload config3 > $O/bench_config3.json 2>> $O/err.txt
for w in config4 config5; do timeout 600 python bench.py --workload $w --docs 2500000 > $O/bench_${w}_2500k.json 2>> $O/err.txt; done
bash tools/gpu_pmc2.sh c4b "--workload config4 --docs 600000" "1 2 4 6 7 8"
python - <<'PY' > $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r2e/*.json")):
    try:
        r = json.load(open(f))
        print("%-28s %7.1f M docs/s  tok %.3f ms  prep %.3f  total %.3f  verified %d  model %s cpu %s" % (f.split("/")[-1], r["value"] / 1e6, r["kernel_ms"]["tokenise"], r["kernel_ms"]["prep"], r["kernel_ms"]["total"], r["verified_docs"], r["config"]["model_file"], r.get("cpu_baseline", {}).get("value")))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/summary.txt; tail -3 $O/pytest_gpu.log; cat $O/smoke.log | tail -2; tail -3 $O/bench_default.err
