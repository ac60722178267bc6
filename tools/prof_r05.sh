#!/bin/bash
# Round-5 profile set on the GPU box (via gpurun), final sources: the default bench (stats, SQ / FETCH / WRITE passes +
# calibration), the same set for the indel-heavy leg and for every other shipped model family (round 4 took kernel stats only for
# those), the MT worker set's kernel stats, then the default bench line itself.
#   tools/prof_r05.sh <tag>    -> gpurun_out/<tag>_*; condense here: for t in a indel hiseq nextseq miseq miseq-legacy; do
#                                  python tools/make_profile_summary.py gpurun_out/<tag>_$t <tag>_$t; done
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
tools/gpu_profile.sh ${TAG}_a > gpurun_out/${TAG}_a.log 2>&1
tools/prof_model.sh ${TAG}_indel "--indel 0.001 0.003" > gpurun_out/${TAG}_indel.log 2>&1
for m in hiseq nextseq miseq miseq-legacy; do
  tools/prof_model.sh ${TAG}_$m "--model $m" 6 > gpurun_out/${TAG}_$m.log 2>&1
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_mtset/stats -o stats --output-format csv -- python $GRAFT_REPO_ROOT/tools/mt_workers_speed.py novaseq 64 256 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_mtset.log 2>&1)
(time python bench.py) > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
tail -c 400 gpurun_out/bench_${TAG}.json; tail -4 gpurun_out/bench_${TAG}.err
