#!/bin/bash
# instruction counts of the wave kernel's phases by difference: the full program, without walking (cfg 14), without walking and moving (cfg 15)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3h; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for v in 3 3587 3843; do
  rm -rf /tmp/prof_r3h
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d /tmp/prof_r3h/pmc_1 -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 --docs 2500000 --variant $v > $O/pmc_$v.log 2>&1
  python $root/tools/prof_summary.py /tmp/prof_r3h $O/pmc_$v.txt > /dev/null 2> $O/summary.err
  echo "variant $v"; grep "k_wp_wave" $O/pmc_$v.txt | cut -c30-140 | grep "VALU\|SALU\|INSTS_LDS\|VMEM"
done
rm -f $O/*.log
