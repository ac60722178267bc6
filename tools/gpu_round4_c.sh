cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
export ISS_SETUP_AHEAD=0
for L in abl1 abl4; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/abl_$L; mkdir -p $OUT
  ISS_MI355X_LIB=$GRAFT_REPO_ROOT/build_ab/libiss_$L.so timeout 200 rocprofv3 --kernel-trace --stats -d $OUT -o stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --indel 0.001 0.003 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads > $OUT/log.txt 2>&1
  echo "== $L"; grep -E "k_indel_script|k_main|k_indel_scan" $OUT/stats_kernel_stats.csv | cut -d, -f1,4 | sed 's/(iss::DevModel.*)"//' 
done
