#!/bin/bash
# Rebuilds a BERT WordPiece lexer model (bert_base_tok / bert_base_cased_tok) from the reference's own sources with the reference's
# own toolchain, two stages of which are replaced by output-identical accelerated versions (see README.md in this directory).
#
#   usage: tools/modelbuild/build_bert_lexer.sh <model> <scratch dir>
#   needs: BF_BUILD   = a build of the reference's tools (fa_* binaries + libfsaCompile.a / libfsaClient.a; cmake/ninja build of /root/reference)
#          BF_SCRIPTS = a copy of the reference's scripts/ with the perl >= 5.30 patch of fa_preproc
#          BF_LDBSRC  = a writable copy of the reference's ldbsrc/
# Runs ~45 minutes (fa_nfalist2nfa ~15, fa_dfa2mindfa ~28), single-threaded, ~8 GB RSS; result: <scratch>/ldb/<model>.bin
set -euo pipefail
m=$1; W=$2
BF_BUILD=${BF_BUILD:-/tmp/bf_build}; BF_SCRIPTS=${BF_SCRIPTS:-/tmp/bf_scripts}; BF_LDBSRC=${BF_LDBSRC:-/tmp/bf_ldb/ldbsrc}; REF=${BF_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
export PATH=$BF_BUILD:$BF_SCRIPTS:$PATH
mkdir -p "$W/ldb" "$W/$m/tmp"; cp -r "$BF_LDBSRC/common" "$W/"; cp "$BF_LDBSRC/Makefile.gnu" "$W/"; cp "$BF_LDBSRC/$m"/*.* "$W/$m/"
cd "$W"; export DICTS_ROOT=.
# helpers (ours): drop-in fa_fsm2fsm_iwec, and fa_fsm2fsm_pack = the reference's main() + library with one class re-implemented
g++ -O2 -o fast_iwec "$HERE/fast_iwec.cpp"
g++ -std=c++11 -O2 -DNDEBUG -DHAVE_ICONV_LIB -DHAVE_NO_SPECSTRINGS -D_VERBOSE -DBLING_FIRE_NOAP -DBLING_FIRE_NOWINDOWS \
    -I $REF/blingfireclient.library/inc -I $REF/blingfireclient.library/src -I $REF/blingfirecompile.library/inc -I $REF/blingfirecompile.library/src \
    "$HERE/FACalcIwEqClasses_fast.cpp" $REF/blingfiretools/fa_fsm2fsm_pack/fa_fsm2fsm_pack.cpp $BF_BUILD/libfsaCompile.a $BF_BUILD/libfsaClient.a -o fa_fsm2fsm_pack_fast
# 1. what scripts/fa_build_lex does (scripts/fa_build_lex:198-229), with fast_iwec in the place of fa_fsm2fsm_iwec
fa_preproc --act-out=$m.act.txt < $m/wbd.lex.utf8 > $m.pre.txt
fa_pr2wre --input-enc=UTF-8 --tagset=$m/wbd.tagset.txt --in=$m.pre.txt --in-actions=$m.act.txt --label=char --out-actions=$m/tmp/wbd.rules.map.txt \
 | fa_re2re_simplify --input-enc=UTF-8 --label=char | fa_re2nfa --input-enc=UTF-8 --label=char --keep-pos \
 | fa_nfalist2nfa --epsilon=3 --nfa-num-base=1114112 \
 | ./fast_iwec --fsm-type=rs-nfa --iw-base=3 --iw-max=1114111 --new-iw-base=3 --out-map=$m/tmp/wbd.rules.fsa.iwmap.txt \
 | fa_nfa2dfa --spec-any=0 | fa_fsm_renum --fsm-type=rs-dfa | fa_dfa2mindfa | fa_fsm_renum --fsm-type=rs-dfa --alg=remove-gaps \
 | fa_fsm2fsm --out=$m/tmp/wbd.rules.fsa.txt --in-type=rs-dfa --out-type=moore-dfa --ow-base=1114112 --ow-max=1073741823
# 2. the Makefile's pack step for the automaton (Makefile.gnu:232-233) with the accelerated equivalence-class stage
./fa_fsm2fsm_pack_fast --alg=triv --type=moore-dfa --remap-iws --use-iwia --in=$m/tmp/wbd.rules.fsa.txt --iw-map=$m/tmp/wbd.rules.fsa.iwmap.txt --out=$m/tmp/wbd.fsa.small.dump
# 3. everything else (configuration, action map, charmap, merge) by the reference's Makefile, unchanged
make -f Makefile.gnu lang=$m all
ls -la ldb/$m.bin
