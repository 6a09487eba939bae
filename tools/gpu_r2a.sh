#!/bin/bash
# round 2, GPU session A: parity + first measurements of the fast-forward lexer, the LDS-table lexer (TextToWords) and UniLane
set -u
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
O=gpurun_out/r2a
{ echo "== $(date) host: $(nproc) cores"; grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -6; } > $O/box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
# headline, the metric's corpus (10 M docs) -- full check, three timings, CPU baseline
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "exit $?" >> $O/bench_default.err
# A/B on the round-1 shard (1.25 M docs): fast-forward on / off
for v in 3 4; do timeout 300 python bench.py --docs 1250000 --variant $v --no-cpu-baseline --no-extra-timings --verify 20000 > $O/ab_lex_v$v.json 2>> $O/ab.err; done
# Unigram: new lane program (unroll 3 / 2 / 4 via tune bits) vs the round-1 ring kernel (variant 3)
for v in 3 6 515 1027; do timeout 300 python bench.py --workload config4 --docs 1250000 --variant $v --no-cpu-baseline --no-extra-timings --verify 20000 > $O/ab_uni_c4_v$v.json 2>> $O/ab.err; done
for v in 3 6; do timeout 300 python bench.py --workload config5 --docs 1250000 --variant $v --no-cpu-baseline --no-extra-timings --verify 20000 > $O/ab_uni_c5_v$v.json 2>> $O/ab.err; done
timeout 300 python bench.py --workload config3 --no-cpu-baseline --no-extra-timings --verify 20000 > $O/c3.json 2>> $O/ab.err
# TextToWords: LDS-resident table vs global table
timeout 200 python tools/bench_words.py 1000000 > $O/words_lds.txt 2>&1
BF_LEX_VARIANT=5 timeout 200 python tools/bench_words.py 1000000 > $O/words_global.txt 2>&1
timeout 100 python tools/bench_words.py 10000 >> $O/words_lds.txt 2>&1
ls -la $O
