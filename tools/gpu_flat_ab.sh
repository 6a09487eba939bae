#!/bin/bash
# the flat program against the wave program on the metric's corpus: parity of a mixed sample through the C-ABI, then the bench line of each
# (verification of 1 M documents against the compiled reference), then the kernel trace of a 2.5 M-document run
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/flat_ab; mkdir -p $O
root=${GRAFT_REPO_ROOT:-$PWD}
timeout 300 python - > $O/parity.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bfutil, blingfire_amd as bf
for model in ["bert_base_tok.bin", "bert_base_cased_tok.bin", "bert_chinese.bin"]:
    mp = bfutil.model_path(model)
    h = bf.load_model(mp)
    ck = bfutil.reference() if bfutil.have_ref() else bfutil.oracle(); hc = ck.load(mp)
    sets = {"adv+fuzz": bf.pack_docs(list(bfutil.ADVERSARIAL) + bfutil.fuzz_docs(3000, seed=11)), "headline": bfutil.gen_workload("headline512", 20000), "config2": bfutil.gen_workload("config2", 30000)}
    for name, (text, off) in sets.items():
        gids, goff = ck.batch(hc, text, off, 512, 100)
        for variant in (4, 5, 3):
            bf.lib().BfSetVariant(h, variant)
            ids, ido = bf.text_to_ids_batch(h, (text, off), 512, 100)
            ok = np.array_equal(ido, goff) and np.array_equal(ids, gids)
            bad = [d for d in range(len(off) - 1) if not np.array_equal(ids[ido[d]:ido[d + 1]], gids[goff[d]:goff[d + 1]])][:3] if not ok else []
            print(model, name, "variant", variant, "OK" if ok else "MISMATCH %s" % bad, "status", bf.lib().BfLastStatus(h), flush=True)
    bf.free_model(h)
PY
cat $O/parity.txt
for v in -1 5; do
  timeout 400 python bench.py --no-cpu-baseline --no-extra-timings --verify 1000000 --steps 5 --warmup 2 --variant $v > $O/bench_v$v.json 2> $O/bench_v$v.err
  python - $O/bench_v$v.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.2f" % j["ms_per_step"], "verified", j.get("verified_docs"), "status", j.get("status"), j.get("kernel_ms"), j["roofline"].get("kernel"), "frac %.4f" % j["roofline"]["frac"])
except Exception as e: print("bench failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-2000:])
PY
done
timeout 300 python bench.py --no-cpu-baseline --no-extra-timings --verify 200000 --steps 5 --warmup 2 --workload config2 > $O/bench_c2.json 2> $O/bench_c2.err
python - $O/bench_c2.json <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); print("config2", "value %.1f M/s" % (j["value"] / 1e6), "ms/step %.3f" % j["ms_per_step"], "verified", j.get("verified_docs"), j.get("kernel_ms"))
except Exception as e: print("bench failed", e)
PY
cd /tmp; rm -rf /tmp/fl_tr
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fl_tr -o tr -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 3 --warmup 1 --docs 2500000 > /dev/null 2> $O/trace.err
f=$(find /tmp/fl_tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f" | cut -c1-220 > $O/kernel_stats.txt; cat $O/kernel_stats.txt
rm -rf /tmp/fl_pmc
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY --kernel-trace -d /tmp/fl_pmc -o pmc -- python $root/bench.py --no-cpu-baseline --no-extra-timings --verify 0 --steps 2 --warmup 1 --docs 2500000 > /dev/null 2> $O/pmc.err
python - /tmp/fl_pmc > $O/pmc.txt 2>&1 <<'PY'
import glob, os, sqlite3, sys
try:
    db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t.startswith("counters_collection")][0]
    for k, c, a in db.execute("select kernel_name, counter_name, avg(value) from %s where kernel_name like '%%k_wp_%%' group by kernel_name, counter_name" % v):
        print(k[:40], c, "%.4g" % a, "(per document %.1f)" % (a / 2.5e6))
except Exception as e: print("pmc failed", e)
PY
cat $O/pmc.txt
