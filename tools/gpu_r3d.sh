#!/bin/bash
# PMC passes (instruction mix, wave-cycle split) of the wave kernel on 2.5 M documents of the headline corpus; one counter group per run
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3d; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
V=${1:--1}
cd /tmp
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_r3d/pmc_$i -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 --docs 2500000 --variant $V > $O/pmc$i.log 2>&1
done
cd $root
python tools/prof_summary.py /tmp/prof_r3d $O/pmc_$V.txt > /dev/null 2> $O/summary.err
rm -f $O/pmc*.log
grep "k_wp_wave" $O/pmc_$V.txt | cut -c30-140 | head -40
