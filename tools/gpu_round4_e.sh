cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_mt_compat.py -x -q -k "bench or short_record_first" > gpurun_out/t2.log 2>&1; echo "pytest rc $?" >> gpurun_out/t2.log
tail -30 gpurun_out/t2.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/b_full.json 2> gpurun_out/b_full.err; echo "bench rc $?"; tail -5 gpurun_out/b_full.err
