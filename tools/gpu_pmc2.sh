#!/bin/bash
# PMC passes for one bench command (each --pmc set in its own rocprofv3 run, no tracing domains besides the kernel trace):
#   tools/gpu_pmc2.sh <tag> "<bench args>"
# writes gpurun_out/pmc2_<tag>/summary.txt (per kernel: every counter averaged over the dispatches, + avg duration)
set -u
tag=$1; args=$2; passes=${3:-all}
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc2_$tag
mkdir -p "$out"
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" \
         "SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  if [ "$passes" != "all" ] && ! echo " $passes " | grep -q " $i "; then continue; fi
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d "$out/p$i" -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 3 --warmup 1 $args > "$out/p$i.log" 2>&1
done
cd "$root"
python - "$out" <<'PY' > "$out/summary.txt"
import glob, sqlite3, sys
rows = {}
for db in sorted(glob.glob(sys.argv[1] + "/p*/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    try:
        q = c.execute("select kernel_name, counter_name, avg(value), avg(duration), count(*) from counters_collection group by kernel_name, counter_name")
    except Exception as e:
        print("skip", db, e); continue
    for k, n, v, d, cnt in q:
        rows.setdefault(k.split("(")[0], {})[n] = (v, d, cnt)
for k in sorted(rows, key=lambda k: -max(v[1] for v in rows[k].values())):
    d = max(v[1] for v in rows[k].values())
    if d < 20000: continue
    print("%s   avg %.3f ms (while counting)" % (k[-110:], d / 1e6))
    for n in sorted(rows[k]):
        print("    %-40s %18.0f   (%d dispatches)" % (n, rows[k][n][0], rows[k][n][2]))
PY
rm -rf "$out"/p[0-9]*     # only summary.txt travels back (gpurun_out/ is capped at 64 MiB)
