#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mt_compat.py -x -q -p no:cacheprovider > $O/t_mt.log 2>&1
grep -v WARNING $O/t_mt.log | tail -4
timeout 300 python tools/mt_workers_speed.py novaseq 8 64 256 > $O/mt_speed.log 2> $O/mt_speed.err
grep "one worker\|\"value\"\|\"workers\"\|per_worker\|error" $O/mt_speed.log | tail -18
timeout 200 python tools/mt_context_probe2.py --torch > $O/ctx_b.log 2>&1; cat $O/ctx_b.log | tail -5
timeout 300 python tools/mt_workers_speed.py hiseq 64 256 > $O/mt_speed_hiseq.log 2>&1
grep "\"value\"\|\"workers\"\|per_worker\|error" $O/mt_speed_hiseq.log | tail -10
timeout 300 python tools/mt_workers_speed.py miseq 256 > $O/mt_speed_miseq.log 2>&1
grep "\"value\"\|\"workers\"\|per_worker\|error" $O/mt_speed_miseq.log | tail -6
