#!/bin/bash
# counters of the _sp kernels (prologue, segmenter, compaction) on 2.5 M documents of config 4 / 5: usage tools/gpu_sp_pmc.sh <workload> [variant]
set -u
export TMPDIR=/tmp
W=${1:-config4}; V=${2:-3}
O=$PWD/gpurun_out/sp_pmc; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
rm -f $O/pmc_${W}_v$V.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE WRITE_SIZE"; do
  rm -rf /tmp/sp_pmc
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/sp_pmc -o pmc -- python $root/bench.py --workload $W --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 --docs ${DOCS:-2500000} --variant $V > /dev/null 2>> $O/pmc.err
  python - /tmp/sp_pmc >> $O/pmc_${W}_v$V.txt 2>&1 <<'PY'
import glob, os, sqlite3, sys
try:
    db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t.startswith("counters_collection")][0]
    for k, c, a in db.execute("select kernel_name, counter_name, avg(value) from %s group by kernel_name, counter_name" % v):
        if any(x in k for x in ("k_prep_sp", "k_uni", "k_seg_unigram", "k_compact", "k_bpe")): print(k[:40], c, "%.4g" % a, "(per document %.1f)" % (a / float(__import__("os").environ.get("DOCS", "2500000"))))
except Exception as e: print("pmc failed", e)
PY
done
cat $O/pmc_${W}_v$V.txt
