#!/bin/bash
# round 3, session E: batched wave program (fill / units / retire loops): parity, A/B of configurations, counters
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3e; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 500 python -m pytest tests/test_gpu_parity_wp.py tests/test_gpu_api.py tests/test_gpu_large_docs.py -m gpu -x -q > $O/pytest_wp.txt 2>&1; tail -3 $O/pytest_wp.txt
Q="--no-cpu-baseline --no-extra-timings --steps 5 --warmup 2"
timeout 400 python bench.py $Q > $O/default_verified.json 2> $O/default_verified.err; python - $O/default_verified.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print("default: verified", j.get("verified_docs"), "ms/step %.2f" % j["ms_per_step"], j["kernel_ms"], "status", j.get("status"))
PY
for v in 259 515 771 1027 2; do
  timeout 300 python bench.py $Q --verify 0 --variant $v > $O/v_$v.json 2> $O/v_$v.err
  python - $O/v_$v.json $v <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print("variant", sys.argv[2], "ms/step %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernel_ms"].items()})
except Exception as e: print("variant", sys.argv[2], "failed", e)
PY
done
timeout 200 python tools/wave_stats.py 1000000 > $O/stats.txt 2>&1; tail -1 $O/stats.txt
cd /tmp
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_r3e/pmc_$i -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 --docs 2500000 > $O/pmc$i.log 2>&1
done
cd $root
python tools/prof_summary.py /tmp/prof_r3e $O/pmc.txt > /dev/null 2> $O/summary.err
rm -f $O/pmc*.log
grep "k_wp_wave" $O/pmc.txt | cut -c30-140 | head -40
