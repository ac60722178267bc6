#!/bin/bash
# Round 5, GPU call 2: the W-workers-per-launch MT mode (tests, rates), world-8 dry run of bench.py, the whole suite.
cd $GRAFT_REPO_ROOT
O=gpurun_out/c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mt_compat.py -x -q -p no:cacheprovider -k "worker_set or cpus8 or cli" > $O/t_mtset.log 2>&1
tail -15 $O/t_mtset.log
timeout 300 python tools/mt_workers_speed.py novaseq > $O/mt_speed.log 2>&1
tail -80 $O/mt_speed.log
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/t_all.log 2>&1
tail -15 $O/t_all.log
