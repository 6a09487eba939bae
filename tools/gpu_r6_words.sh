#!/bin/bash
# round 6, the words modes: GPU tests of tests/test_words.py (long documents among them), the config 1 line and its kernel trace
set -u
tag=${1:-r06_words}; root=$PWD; O=$PWD/gpurun_out/$tag; mkdir -p $O
if [ "${2:-}" != "notest" ]; then timeout 900 python -m pytest tests/test_words.py -m gpu -x -q > $O/pytest_words.txt 2>&1; tail -5 $O/pytest_words.txt; fi
timeout 300 python bench.py --workload config1 > $O/bench_config1.json 2> $O/bench_config1.err; cut -c1-330 $O/bench_config1.json; tail -3 $O/bench_config1.err
P=/tmp/prof_words; rm -rf $P
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $P/stats -o stats -- python $root/bench.py --workload config1 --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_config1_traced.json 2> /dev/null
cd $root
python tools/prof_summary.py $P $O/config1_kernels.txt > /dev/null 2> $O/summary.err; head -30 $O/config1_kernels.txt | cut -c1-180
