cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hip_matches_oracle or both_indel_paths or scripted or bam_built or tuning or back_to_back or counter_rings or store_mutations or chunked or randomized_differential or more_work_items" > gpurun_out/t1.log 2>&1; echo "pytest rc $?" >> gpurun_out/t1.log
tail -25 gpurun_out/t1.log
timeout 300 python bench.py --steps 10 --warmup 3 --indel 0.001 0.003 --no-cpu-baseline --no-end-to-end --no-other-workloads > gpurun_out/b_indel.json 2> gpurun_out/b_indel.err; echo "bench indel rc $?"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_indel_a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --indel 0.001 0.003 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads"
ISS_SETUP_AHEAD=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $BENCH > $OUT/stats.log 2>&1
head -12 $OUT/stats/stats_kernel_stats.csv | cut -c1-160
