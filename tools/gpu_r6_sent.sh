#!/bin/bash
# round 6: TextToSentences on one 1 MB document against the same bytes as short documents, with the kernel trace of each case (the traces: the text without the lines that hold a run of ten spaces)
set -u
tag=${1:-r06_sent}; root=$PWD; O=$PWD/gpurun_out/$tag; mkdir -p $O
timeout 300 python tools/bench_sentences.py 2>&1 | grep -v "amdgpu.ids\|Warn\|dt, do" | tee $O/sentences.txt
# the same without the 44 lines of the file that hold a run of ten spaces (one start position of such a line: 6,440 sequential steps under sbd.bin)
timeout 300 python tools/bench_sentences.py 1048576 0 -1 noruns 2>&1 | grep -v "amdgpu.ids\|Warn\|dt, do" | tee $O/sentences_noruns.txt
timeout 300 python tools/bench_sentences.py 1048576 0x40000000 -1 noruns 2>&1 | grep -v "amdgpu.ids\|Warn\|dt, do" | tee -a $O/sentences_noruns.txt
for case in 0 1 2; do
  P=/tmp/prof_sent$case; rm -rf $P
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $P/stats -o stats -- python $root/tools/bench_sentences.py 1048576 0 $case noruns > /dev/null 2>&1
  cd $root
  python tools/prof_summary.py $P $O/sentences_kernels_case$case.txt > /dev/null 2> $O/summary.err; echo "case $case"; sed -n 2,12p $O/sentences_kernels_case$case.txt | cut -c1-120
done
