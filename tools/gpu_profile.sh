#!/bin/bash
# rocprofv3 passes for the bench command (kernel trace/stats, then separate PMC passes as the guide prescribes).
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --verify 0 $*"
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/stats" -o stats -- $B > "$out/stats.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" -o pmc -- $B > "$out/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" -o pmc -- $B > "$out/pmc_write.log" 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$out/pmc_tcc" -o pmc -- $B > "$out/pmc_tcc.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$out/pmc_sq" -o pmc -- $B > "$out/pmc_sq.log" 2>&1
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum -d "$out/pmc_tcp" -o pmc -- $B > "$out/pmc_tcp.log" 2>&1
timeout 600 rocprofv3 --pmc TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d "$out/pmc_tcc2" -o pmc -- $B > "$out/pmc_tcc2.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d "$out/pmc_sq2" -o pmc -- $B > "$out/pmc_sq2.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find "$out" -name "*.csv" | head -50 > "$out/files.txt"
# keep the merged payload small: drop per-dispatch traces bigger than 8 MB
find "$out" -size +8M -delete
