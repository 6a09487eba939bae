#!/bin/bash
# the BPE wave program on the GPU (variant bit 0x40): parity test, then config 3 with and without it
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/bpe_wave; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_sp.py -m gpu -x -q -k "bpe_wave_program" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
Q="--no-cpu-baseline --no-extra-timings --steps 5 --warmup 2 --workload config3"
for v in 0 64; do
  timeout 300 python bench.py $Q --variant $v > $O/config3_v$v.json 2> $O/config3_v$v.err
  python - $O/config3_v$v.json $v <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print("variant", sys.argv[2], "verified", j.get("verified_docs"), "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], {k: round(v, 2) for k, v in j["kernel_ms"].items()}, "status", j.get("status"))
except Exception as e: print("variant", sys.argv[2], "failed", e); print(open(sys.argv[1].replace('.json', '.err')).read()[-600:])
PY
done
timeout 200 python tools/bpe_wave_stats.py 200000 2>&1 | tail -1
# kernel split of the variant (rocprofv3 --kernel-trace --stats)
cd /tmp; rm -rf /tmp/prof_bw
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_bw/stats -o stats -- python $OLDPWD/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 3 --warmup 1 --workload config3 --variant 64 > /dev/null 2> $O/prof.err
cd $OLDPWD
python tools/prof_summary.py /tmp/prof_bw $O/prof.txt > /dev/null 2>> $O/prof.err; head -14 $O/prof.txt | cut -c1-120
