#!/bin/bash
# round 6, VERDICT item 6: k_wp_merge<true> (the offsets line's merge) has two modes, one per process.  Processes of the offsets line under
# rocprofv3 --pmc restricted to that kernel: its duration and counters per process.  usage: gpu_r6_merge_modes.sh <tag> <tcp|tcc> <processes>
set -u
tag=${1:-r06_merge_modes}; set_=${2:-tcc}; np=${3:-4}; root=$PWD; O=$PWD/gpurun_out/$tag; mkdir -p $O
B="python $root/bench.py --offsets --verify 0 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-timings"
if [ $set_ = tcp ]; then C="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
else C="TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"; fi
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $np); do
  rm -rf /tmp/mm; t0=$(date +%s)
  timeout 240 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "k_wp_merge" -d /tmp/mm -o pmc -- $B > /dev/null 2> /tmp/mm.err
  echo -n "$set_$i ($(( $(date +%s) - t0 )) s) "; python $root/tools/merge_modes.py /tmp/mm 2>&1 | tail -1
done | tee $O/merge_modes_$set_.txt
tail -3 /tmp/mm.err | cut -c1-200
