cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in "" "--model hiseq"; do echo "== $m"; bash tools/ab_multi.sh "$m" build_ab/libiss_base.so build_ab/libiss_nt.so 2>&1 | tee -a gpurun_out/ab_nt.log; done
