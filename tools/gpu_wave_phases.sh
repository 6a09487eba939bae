#!/bin/bash
# instruction counts of k_wp_wave variants (phase costs by difference): tools/gpu_wave_phases.sh "<variants>"   (0 = shipped)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/wave_phases; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for v in ${1:-0 3584 3840}; do
  rm -rf /tmp/prof_wave_phases_$v
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --kernel-trace -d /tmp/prof_wave_phases_$v/pmc_1 -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 --docs 2500000 --variant $v > $O/v$v.log 2>&1
  python $root/tools/prof_summary.py /tmp/prof_wave_phases_$v $O/pmc_$v.txt > /dev/null 2>> $O/summary.err
  echo "variant $v"; grep "k_wp_wave" $O/pmc_$v.txt | grep "SQ_" | cut -c46-130
done
rm -f $O/v*.log
