#!/bin/bash
# round 2, GPU session B: new tests, kernel-trace profile of the default bench, PMC passes (lexer, Unigram x2, BPE)
set -u
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
root=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $root/$O/trace_default -o t -- python $root/bench.py > $root/$O/bench_default_traced.json 2> $root/$O/bench_default_traced.err )
find $O/trace_default -name "*kernel_stats*" | head -3
bash tools/gpu_pmc2.sh lex "--docs 1250000"
bash tools/gpu_pmc2.sh c4 "--workload config4 --docs 1250000"
bash tools/gpu_pmc2.sh c5 "--workload config5 --docs 1250000"
bash tools/gpu_pmc2.sh c3 "--workload config3"
for w in config4 config5; do timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
find $O -size +6M -delete
ls -la $O gpurun_out/pmc2_*
