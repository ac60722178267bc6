#!/bin/bash
# Round 5, GPU call 4: worker set with the emitter / walker beside the resolver and prefetch of the missing words only.
cd $GRAFT_REPO_ROOT
O=gpurun_out/c4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mt_compat.py -x -q -p no:cacheprovider > $O/t_mt.log 2>&1
grep -v WARNING $O/t_mt.log | tail -8
timeout 300 python tools/mt_workers_speed.py novaseq 1 8 64 256 > $O/mt_speed.log 2>&1
grep "one worker\|\"value\"\|\"workers\"\|per_worker\|error" $O/mt_speed.log | tail -40
ISS_MT_SET_TURN=1024 timeout 300 python tools/mt_workers_speed.py novaseq 256 > $O/mt_speed_1024.log 2>&1
grep "\"value\"\|\"workers\"\|per_worker\|error" $O/mt_speed_1024.log | tail -8
ISS_MT_SET_TURN=4096 timeout 300 python tools/mt_workers_speed.py novaseq 64 > $O/mt_speed_4096.log 2>&1
grep "\"value\"\|\"workers\"\|per_worker\|error" $O/mt_speed_4096.log | tail -8
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o mt --output-format csv -- python $GRAFT_REPO_ROOT/tools/mt_workers_speed.py novaseq 256 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
head -9 $O/prof/*kernel_stats.csv | cut -c1-150
