#!/bin/bash
# Library builds with compile-time switches for interleaved A/B runs on the GPU box (tools/ab_multi.sh, tools/ab_models.sh):
#   tools/build_variants.sh name1 "-DISS_STAGE=1" name2 "-DISS_STAGE=2 -DISS_SWP=1" ...   -> build_ab/libiss_<name>.so
# (build_ab/ is git-ignored but travels with gpurun's snapshot)
mkdir -p build_ab
ID=$(python -c "import __graft_entry__ as g; print(g.source_hash())")
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DISS_BUILD_ID="\"$ID-$NAME\"" $FLAGS -I include \
      -o build_ab/libiss_$NAME.so insilicoseq_amd/csrc/iss_mi355x.hip &
done
wait
ls -la build_ab/
