cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in "" "--model hiseq"; do echo "== $m"; bash tools/ab_multi.sh "$m" build_ab/libiss_x0.so build_ab/libiss_x3.so build_ab/libiss_x4.so build_ab/libiss_x5.so 2>&1 | tee -a gpurun_out/ab_x.log; done
