cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/t_full.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/t_full.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-end-to-end --no-cpu-baseline > gpurun_out/b_g.json 2> gpurun_out/b_g.err; echo "bench rc $?"
