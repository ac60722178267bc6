#!/bin/bash
# Round 5, GPU call 7: screening fix (fewer pairs to the walker), and where bench.py's MT leg loses 40 %.
cd $GRAFT_REPO_ROOT
O=gpurun_out/c7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mt_compat.py -x -q -p no:cacheprovider > $O/t_mt.log 2>&1
grep -v WARNING $O/t_mt.log | tail -4
ISS_MT_SET_DEBUG=1 timeout 300 python tools/mt_workers_speed.py novaseq 8 64 256 > $O/mt_speed.log 2> $O/mt_speed.err
grep "\"value\"\|\"workers\"\|per_worker\|error" $O/mt_speed.log | tail -16
grep "mt set" $O/mt_speed.err | awk 'NR%5==1' | head -12
timeout 200 python tools/mt_context_probe2.py > $O/ctx_a.log 2>&1; cat $O/ctx_a.log
timeout 200 python tools/mt_context_probe2.py --torch > $O/ctx_b.log 2>&1; cat $O/ctx_b.log
GPU_MAX_HW_QUEUES=8 timeout 200 python tools/mt_context_probe2.py --torch > $O/ctx_c.log 2>&1; echo "GPU_MAX_HW_QUEUES=8:"; cat $O/ctx_c.log
