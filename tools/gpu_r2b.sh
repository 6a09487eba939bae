#!/bin/bash
# round 2, GPU session B: new tests, kernel-trace profile of the default bench, PMC passes (lexer, Unigram); only small text
# summaries are left under gpurun_out/ (it is capped at 64 MiB)
set -u
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
root=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/trace_default -o t -- python $root/bench.py > $root/$O/bench_default_traced.json 2> $root/$O/bench_default_traced.err )
for f in $(find /tmp/trace_default -name "*kernel_stats.csv" -o -name "*domain_stats.csv"); do cp $f $O/; done
bash tools/gpu_pmc2.sh lex "--docs 600000"
bash tools/gpu_pmc2.sh c4 "--workload config4 --docs 600000"
bash tools/gpu_pmc2.sh c3 "--workload config3 --docs 300000" "1 2 4 6"
for w in config4 config5; do timeout 600 python bench.py --workload $w --docs 2500000 > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 300 python tools/bench_single_calls.py > $O/single_calls.txt 2>&1
du -sh gpurun_out
