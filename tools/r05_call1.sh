#!/bin/bash
# Round 5, GPU call 1: staged lookups (ISS_STAGE / ISS_SWP builds) against the base, interleaved; the suite on the most changed build.
cd $GRAFT_REPO_ROOT
O=gpurun_out/c1; mkdir -p $O
L="build_ab/libiss_base.so build_ab/libiss_s1.so build_ab/libiss_s2.so build_ab/libiss_s1w.so build_ab/libiss_s2w.so"
tools/ab_multi.sh "" $L > $O/ab_nova.log 2>&1
tools/ab_multi.sh "--model hiseq" $L > $O/ab_hiseq.log 2>&1
tools/ab_multi.sh "--model miseq" build_ab/libiss_base.so build_ab/libiss_s2.so build_ab/libiss_s2w.so > $O/ab_miseq.log 2>&1
tools/ab_multi.sh "--model nextseq" build_ab/libiss_base.so build_ab/libiss_s2w.so > $O/ab_nextseq.log 2>&1
ISS_MI355X_LIB=$PWD/build_ab/libiss_s2w.so timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/t_s2w.log 2>&1
tail -3 $O/t_s2w.log
cat $O/ab_*.log
