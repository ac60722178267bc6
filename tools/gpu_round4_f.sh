cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in "--indel 0.001 0.003"; do echo "== $m"; bash tools/ab_multi.sh "$m" build_ab/libiss_new.so build_ab/libiss_occ5.so build_ab/libiss_co.so 2>&1 | tee -a gpurun_out/ab_f.log; done
