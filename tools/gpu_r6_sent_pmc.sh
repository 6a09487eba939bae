cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ps; timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --kernel-include-regex "k_lex_long<" -d /tmp/ps -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_sentences.py 1048576 0 0 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/merge_modes.py /tmp/ps "k_lex_long<true, false, false>" | cut -c1-600
python $GRAFT_REPO_ROOT/tools/merge_modes.py /tmp/ps "k_lex_long<true, false, true>" | cut -c1-600
