#!/bin/bash
# Round-2 closing evidence (after the lexer / host-path changes of the second half of the round): GPU test tier, smoke, the default
# bench line under `rocprofv3 --kernel-trace --stats` plus FETCH_SIZE / WRITE_SIZE passes, then the bench lines of the other configs.
# Only text summaries are left under gpurun_out/ (capped at 64 MiB).
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/final3; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_default/stats -o stats -- python $root/bench.py > $O/default_bench_traced.json 2> $O/default_stats.err
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_default/pmc_$i -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 3 --warmup 1 > $O/default_pmc$i.log 2>&1
done
cd $root
python tools/prof_summary.py /tmp/prof_default $O/default.txt > /dev/null 2> $O/default_summary.err
rm -f $O/default_pmc*.log $O/default_stats.err
timeout 400 python bench.py > $O/default_bench.json 2> /dev/null
for w in config4 config5 config3 config2 config1; do
  timeout 400 python bench.py --workload $w > $O/${w}_bench.json 2> /dev/null
done
du -sh gpurun_out; ls -la $O
