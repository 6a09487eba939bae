#!/bin/bash
# Round-2 closing session: A/B of the lexer's load-time shortcuts (variant 9 = without), the GPU test tier, smoke, the default bench
# line, then `rocprofv3 --kernel-trace --stats` and FETCH/WRITE passes of the default command.  Text summaries only under gpurun_out/.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2h; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
Q="--no-cpu-baseline --no-extra-timings --verify 0 --steps 5 --warmup 2"
timeout 300 python bench.py $Q > $O/ab_new.json 2> $O/ab_new.err
timeout 300 python bench.py $Q --variant 9 > $O/ab_v9.json 2> $O/ab_v9.err
timeout 300 python bench.py $Q --workload config2 > $O/ab_c2_new.json 2>> $O/ab_new.err
timeout 300 python bench.py $Q --workload config2 --variant 9 > $O/ab_c2_v9.json 2>> $O/ab_v9.err
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 600 python bench.py > $O/default_bench.json 2> $O/default_bench.err
timeout 300 python bench.py --workload config2 > $O/config2_bench.json 2> /dev/null
timeout 200 python bench.py --workload config1 > $O/config1_bench.json 2> /dev/null
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_default/stats -o stats -- python $root/bench.py > $O/default_bench_traced.json 2> $O/default_stats.err
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_default/pmc_$i -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 3 --warmup 1 > $O/default_pmc$i.log 2>&1
done
cd $root
python tools/prof_summary.py /tmp/prof_default $O/default.txt > /dev/null 2> $O/default_summary.err
rm -f $O/default_pmc*.log $O/default_stats.err
du -sh gpurun_out; tail -3 $O/pytest_gpu.txt; cat $O/smoke.txt | tail -1; cat $O/ab_new.json $O/ab_v9.json $O/ab_c2_new.json $O/ab_c2_v9.json | cut -c1-200
