#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mt_compat.py -x -q -p no:cacheprovider > $O/t_mt.log 2>&1
grep -v WARNING $O/t_mt.log | tail -4
ISS_MT_SET_DEBUG=1 timeout 300 python tools/mt_workers_speed.py novaseq 8 64 256 > $O/mt_speed.log 2> $O/mt_speed.err
grep "\"value\"\|\"workers\"\|per_worker\|error" $O/mt_speed.log | tail -14
grep "mt set" $O/mt_speed.err | awk 'NR%6==1' | head -8
for off in 1000 2000 3000; do ISS_FUZZ_OFFSET=$off ISS_MT_SET_TURN=$((off/40)) timeout 300 python -m pytest tests/test_gpu_mt_compat.py -x -q -p no:cacheprovider -k "worker_set or cpus8" > $O/t_soak_$off.log 2>&1; grep -v WARNING $O/t_soak_$off.log | tail -2; done
