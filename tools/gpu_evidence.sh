#!/bin/bash
# Evidence of the build in this tree (run it LAST in a round; nothing touches blingfire_amd/csrc afterwards): GPU test tier + smoke, the
# calibration of FETCH_SIZE / WRITE_SIZE for this kernel's streaming access shapes, rocprofv3 --kernel-trace --stats and separate --pmc
# passes of the default command (-> profiles/traffic.json keyed by the csrc hash), the bench lines of every configuration.
# usage: tools/gpu_evidence.sh <tag> [quick]       outputs: gpurun_out/<tag>/ (text summaries only; copy what is to be judged into profiles/)
set -u
export TMPDIR=/tmp
tag=${1:-evidence}; quick=${2:-}      # quick: "quick" = the default command only, "lines" = only the bench lines of configs 1..5 (profiles/traffic.json as it is)
O=$PWD/gpurun_out/$tag; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
git -C $root rev-parse HEAD > $O/head.txt 2>/dev/null
if [ "$quick" = lines ]; then
  for w in config2 config1 config3 config4 config5; do timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; tail -c 200 $O/bench_$w.json; echo; done
  exit 0
fi
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
# ---- calibration: known bytes / counter for 8-byte-per-lane reads (the text read of k_wp_wave), 16-byte reads, 4-byte writes
cd /tmp
rm -rf /tmp/cal; mkdir -p /tmp/cal
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/cal/pmc_fetch -o pmc -- $root/tools/microbench/stream 2048 > $O/cal_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/cal/pmc_write -o pmc -- $root/tools/microbench/stream 2048 > $O/cal_write.log 2>&1
python - /tmp/cal $root/profiles/fetch_calibration.json <<'PY'
import glob, json, os, sqlite3, sys
src, dst = sys.argv[1], sys.argv[2]
bytes_known = 2048 << 20
out = {"known_bytes_per_kernel": bytes_known, "source": "tools/microbench/stream.hip under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB)"}
def val(sub, counter, kern):
    db = sqlite3.connect(glob.glob(os.path.join(src, sub, "*.db"))[0])
    r = db.execute("select avg(value) from counters_collection where kernel_name like ? and counter_name = ?", ("%" + kern + "%", counter)).fetchone()
    return r[0]
try:
    f8, f16, w4 = val("pmc_fetch", "FETCH_SIZE", "k_read<8>"), val("pmc_fetch", "FETCH_SIZE", "k_read<16>"), val("pmc_write", "WRITE_SIZE", "k_write4")
    out.update({"FETCH_SIZE_KiB_read8": f8, "FETCH_SIZE_KiB_read16": f16, "WRITE_SIZE_KiB_write4": w4,
                "read8_bytes_per_counted_byte": bytes_known / (f8 * 1024.0), "read16_bytes_per_counted_byte": bytes_known / (f16 * 1024.0),
                "write4_bytes_per_counted_byte": bytes_known / (w4 * 1024.0)})
except Exception as e:
    out["error"] = str(e)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
PY
rm -f $O/cal_*.log
# ---- the default command: kernel trace + stats, then one counter group per pass
P=/tmp/prof_$tag; rm -rf $P
B="python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $P/stats -o stats -- $B > $O/default_traced.json 2> $O/stats.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/pmc_fetch -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/pmc_write -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $P/pmc_tcc -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $P/pmc_sq1 -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $P/pmc_sq2 -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2>&1
cd $root
python tools/prof_summary.py $P $O/default.txt > /dev/null 2> $O/summary.err
model=$(python -c "import sys; sys.path.insert(0,'tests'); import bfutil; print(bfutil.bert_model_name())")
kern=$(python -c "import json; print(json.load(open('$O/default_traced.json'))['roofline']['kernel'].split('(')[1].rstrip(')'))" 2>/dev/null || echo k_wp_wave)
python tools/prof_traffic.py $P "headline512/$model/10000000" $kern 1 > $O/traffic_default.txt 2>&1; tail -1 $O/traffic_default.txt
rm -f $O/stats.err
# ---- the lines
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
# the offsets API of the metric's corpus (TextToIdsWithOffsetsBatchDevice: the wave program's instance that carries a span with every id + k_compact_text)
timeout 400 python bench.py --offsets --no-cpu-baseline --no-extra-timings --verify 2000000 --steps 5 --warmup 2 > $O/bench_default_offsets.json 2> $O/bench_default_offsets.err; tail -c 200 $O/bench_default_offsets.json; echo
if [ -z "$quick" ]; then
  # traffic of the SentencePiece-style kernels (one sub-batch launch each): before their bench lines, which quote it
  for spec in "config3 gpt2.bin 1000000 k_bpe_wave 1" "config4 xlm_roberta_base.bin 10000000 k_seg_unigram_lane 4" "config5 laser500k.bin 10000000 k_seg_unigram_lane 4"; do
    set -- $spec
    Q=/tmp/prof_${tag}_$1; rm -rf $Q
    cd /tmp
    for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      d=pmc_fetch; [ "$c" = "WRITE_SIZE" ] && d=pmc_write; [ "$c" = "TCC_HIT_sum TCC_MISS_sum" ] && d=pmc_tcc
      timeout 500 rocprofv3 --pmc $c --kernel-trace -d $Q/$d -o pmc -- python $root/bench.py --workload $1 --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 > /dev/null 2>&1
    done
    cd $root
    python tools/prof_traffic.py $Q "$1/$2/$3" $4 $5 > $O/traffic_$1.txt 2>&1; tail -1 $O/traffic_$1.txt
  done
  for w in config2 config1 config3 config4 config5; do
    timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; python - $O/bench_$w.json $w <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"))
except Exception as e: print(sys.argv[2], "failed", e)
PY
  done
fi
cp profiles/traffic.json profiles/fetch_calibration.json $O/ 2>/dev/null
ls $O
