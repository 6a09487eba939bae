#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3f; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
Q="--no-cpu-baseline --no-extra-timings --steps 5 --warmup 2"
timeout 400 python bench.py $Q --verify 2000000 > $O/default_verified.json 2> $O/default_verified.err; python - $O/default_verified.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); print("default: verified", j.get("verified_docs"), "ms/step %.2f" % j["ms_per_step"], j["kernel_ms"], "status", j.get("status"))
PY
for v in ${1:-259 515}; do
  timeout 300 python bench.py $Q --verify 0 --variant $v > $O/v_$v.json 2> $O/v_$v.err
  python - $O/v_$v.json $v <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print("variant", sys.argv[2], "ms/step %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernel_ms"].items()})
except Exception as e: print("variant", sys.argv[2], "failed", e)
PY
done
cd /tmp
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_r3f/pmc_$i -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 --docs 2500000 > $O/pmc$i.log 2>&1
done
cd $root
python tools/prof_summary.py /tmp/prof_r3f $O/pmc.txt > /dev/null 2> $O/summary.err
rm -f $O/pmc*.log
grep "k_wp_wave" $O/pmc.txt | cut -c30-140 | head -40
