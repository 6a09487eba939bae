#!/bin/bash
# round 6: pass 2 of the long-document form under sbd.bin against the size of the one document (latency of one position, or throughput?)
set -u
root=$PWD; O=$PWD/gpurun_out/r06_sent_sizes; mkdir -p $O
for n in 16384 65536 262144 4194304; do
  P=/tmp/prof_ss; rm -rf $P; cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $P/stats -o stats -- python $root/tools/bench_sentences.py $n 0 0 > /dev/null 2>&1
  cd $root; python tools/prof_summary.py $P /tmp/ss.txt > /dev/null 2>&1
  echo "one document of $n bytes:"; grep "k_lex_long<\|k_lex_wp" /tmp/ss.txt | head -3 | cut -c1-110
done | tee $O/sizes.txt
