#!/bin/bash
# Evidence of the build in this tree (run it LAST in a round; nothing touches blingfire_amd/csrc afterwards): GPU test tier + smoke, the
# calibration of FETCH_SIZE / WRITE_SIZE for the kernels' streaming access shapes, and for EVERY bench line (the default, its offsets form,
# configs 2..5): rocprofv3 --kernel-trace --stats -> <name>.txt, separate --pmc passes summed over all kernels of a step ->
# profiles/traffic.json (keyed by workload/model/documents[/offsets] and the csrc hash), then the bench line itself.
# usage: tools/gpu_evidence.sh <tag> [quick|lines]      outputs: gpurun_out/<tag>/ (text summaries only; copy what is to be judged into profiles/)
set -u
export TMPDIR=/tmp
tag=${1:-evidence}; quick=${2:-}      # "quick" = tests + the default command only; "lines" = only the bench lines (profiles/traffic.json as it is)
O=$PWD/gpurun_out/$tag; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
git -C $root rev-parse HEAD > $O/head.txt 2>/dev/null
line() { # name, bench args...: the bench line of a configuration
  local name=$1; shift
  timeout 900 python $root/bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); r = j["roofline"]
    print(sys.argv[2], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"),
          "| %s %.2f ms frac %.4f | step frac %.4f | traffic %s stale %s" % (r["kernel"], r.get("kernel_ms", 0), r["frac"], r.get("step", {}).get("frac", 0), r.get("traffic"), r.get("traffic_stale")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
if [ "$quick" = lines ]; then
  line default; line default_offsets --offsets --verify 2000000; for w in config2 config1 config3 config4 config5; do line $w --workload $w; done
  exit 0
fi
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
# ---- calibration: known bytes / counter for 8-byte-per-lane reads (the text read of the WordPiece kernels), 16-byte reads, 4-byte writes
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
prof() { # name, traffic key, dominant kernel, launches of it per step, more counters (0/1), bench args...
  local name=$1 key=$2 kern=$3 per=$4 more=$5; shift 5
  local P=/tmp/prof_${tag}_$name; rm -rf $P
  local B="python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 $*"
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $P/stats -o stats -- $B > $O/${name}_traced.json 2> /dev/null
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/pmc_fetch -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/pmc_write -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2>&1
  if [ "$more" = 1 ]; then
    timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $P/pmc_tcc -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $P/pmc_sq1 -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $P/pmc_sq2 -o pmc -- $B --steps 3 --warmup 1 > /dev/null 2>&1
  fi
  cd $root
  python tools/prof_summary.py $P $O/$name.txt > /dev/null 2> $O/${name}_summary.err
  python tools/prof_traffic.py $P "$key" $kern $per > $O/traffic_$name.txt 2>&1; head -1 $O/traffic_$name.txt | cut -c1-300
  rm -rf $P
}
model=$(python -c "import sys; sys.path.insert(0,'$root/tests'); import bfutil; print(bfutil.bert_model_name())")
prof default "headline512/$model/10000000" k_wp_flat 1 1
line default
if [ -z "$quick" ]; then
  prof default_offsets "headline512/$model/10000000/offsets" k_wp_flat 1 0 --offsets
  line default_offsets --offsets --verify 2000000
  prof config2 "config2/$model/1000000" k_wp_flat 1 0 --workload config2
  prof config3 "config3/gpt2.bin/1000000" k_bpe_wave 1 0 --workload config3
  prof config4 "config4/xlm_roberta_base.bin/10000000" k_uni_cut 4 0 --workload config4
  prof config5 "config5/laser500k.bin/10000000" k_uni_cut 4 0 --workload config5
  for w in config2 config1 config3 config4 config5; do line $w --workload $w; done
fi
cp profiles/traffic.json profiles/fetch_calibration.json $O/ 2>/dev/null
ls $O
