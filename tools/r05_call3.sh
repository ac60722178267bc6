#!/bin/bash
# Round 5, GPU call 3: the worker set after the move / word-budget fixes: tests, rates, per-kernel times; then the whole suite.
cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mt_compat.py -x -q -p no:cacheprovider -k "worker_set or cpus8" > $O/t_mtset.log 2>&1
grep -v WARNING $O/t_mtset.log | tail -8
timeout 300 python tools/mt_workers_speed.py novaseq 1 8 64 256 > $O/mt_speed.log 2>&1
grep -v "^ *\"\(unit\|calls\|pairs\|seconds\|sample\)" $O/mt_speed.log | tail -60
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o mt --output-format csv -- python $GRAFT_REPO_ROOT/tools/mt_workers_speed.py novaseq 64 256 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
head -12 $O/prof/*kernel_stats.csv | cut -c1-180
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/t_all.log 2>&1
grep -v WARNING $O/t_all.log | tail -8
