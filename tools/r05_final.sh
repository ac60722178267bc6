#!/bin/bash
# Round 5, final GPU call: the round's profile set on the final sources (tools/prof_r05.sh), a captured soak of the randomized
# tests on other configurations, the whole suite.
cd $GRAFT_REPO_ROOT
tools/prof_r05.sh r05
tools/soak.sh 3 150000 > gpurun_out/soak_r05.log 2>&1; tail -2 gpurun_out/soak_r05.log
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/t_r05_all.log 2>&1
grep -v WARNING gpurun_out/t_r05_all.log | tail -4
