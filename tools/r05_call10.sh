#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mt_compat.py -x -q -p no:cacheprovider > $O/t_mt.log 2>&1
grep -v WARNING $O/t_mt.log | tail -4
ISS_MT_SET_DEBUG=1 timeout 300 python tools/mt_workers_speed.py novaseq 8 64 256 > $O/mt_speed.log 2> $O/mt_speed.err
grep "\"value\"\|\"workers\"\|per_worker\|error" $O/mt_speed.log | tail -14
grep "mt set" $O/mt_speed.err | awk 'NR%6==1' | head -8
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_mt_compat.py > $O/t_all.log 2>&1
grep -v WARNING $O/t_all.log | tail -6
