#!/bin/bash
# Round 5, GPU call 6: worker set with three stream buffers (tests + rates + what a turn moves), and the MT leg's context effect.
cd $GRAFT_REPO_ROOT
O=gpurun_out/c6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_mt_compat.py -x -q -p no:cacheprovider -k "worker_set or cpus8" > $O/t_mt.log 2>&1
grep -v WARNING $O/t_mt.log | tail -4
ISS_MT_SET_DEBUG=1 timeout 300 python tools/mt_workers_speed.py novaseq 8 64 256 > $O/mt_speed.log 2> $O/mt_speed.err
grep "\"value\"\|\"workers\"\|per_worker\|error" $O/mt_speed.log | tail -16
grep "mt set" $O/mt_speed.err | awk 'NR%4==1' | head -30
ISS_MT_SET_TURN=1024 timeout 300 python tools/mt_workers_speed.py novaseq 64 > $O/mt_speed_1024.log 2>&1
grep "\"value\"\|per_worker\|error" $O/mt_speed_1024.log | tail -4
timeout 300 python tools/mt_context_probe.py > $O/mt_context.log 2>&1
cat $O/mt_context.log
