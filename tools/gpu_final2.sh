#!/bin/bash
# Round-end evidence for profiles/: for the default bench line and configs 3/4/5 -- `rocprofv3 --kernel-trace --stats` of the same
# command, then FETCH_SIZE / WRITE_SIZE / TCC hit-miss in separate --pmc passes (MI355X_MICROARCH.md: one counter group per run, no
# tracing domains besides the kernel trace).  Only text summaries are left under gpurun_out/ (capped at 64 MiB).
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/final2; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
run_one() {   # tag, bench args, number of PMC passes (first N of the list)
  tag=$1; args=$2; np=${3:-5}
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag/stats -o stats -- python $root/bench.py $args > $O/${tag}_bench_traced.json 2> $O/${tag}_stats.err
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    if [ $i -gt $np ]; then break; fi
    timeout 900 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_$tag/pmc_$i -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 3 --warmup 1 $args > $O/${tag}_pmc$i.log 2>&1
  done
  cd $root
  python tools/prof_summary.py /tmp/prof_$tag $O/${tag}.txt > /dev/null 2> $O/${tag}_summary.err
  rm -f $O/${tag}_pmc*.log $O/${tag}_stats.err
}
run_one default ""
run_one config3 "--workload config3" 3
run_one config4 "--workload config4" 3
run_one config5 "--workload config5" 3
run_one config2 "--workload config2" 0
timeout 200 python bench.py --workload config1 > $O/config1_bench.json 2> /dev/null
du -sh gpurun_out; ls -la $O
