cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "records_of_2_31 or beyond_2_30 or hip_matches_oracle or randomized_differential or custom_fragment or scripted or more_work_items or worker_iterator" > gpurun_out/t3.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/t3.log
