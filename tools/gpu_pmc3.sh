#!/bin/bash
# PMC passes of the default command after the second-half changes (8 waves per SIMD): instruction mix, wave-cycle split, L2 hit rate.
# One counter group per run, kernel trace only (MI355X_MICROARCH.md).
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/pmc3; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_pmc3/pmc_$i -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 3 --warmup 1 > $O/pmc$i.log 2>&1
done
cd $root
python tools/prof_summary.py /tmp/prof_pmc3 $O/default_pmc.txt > /dev/null 2> $O/summary.err
rm -f $O/pmc*.log
grep -A12 "pmc pass" $O/default_pmc.txt | grep "k_lex_wp_plain" 
