#!/bin/bash
# Round-6 profile set on the GPU box (via gpurun), final sources: the default bench (stats, SQ / FETCH / WRITE passes +
# calibration + the GRBM clock pass), the same set for the indel-heavy leg and for every other shipped model family (k_main_g on
# HiSeq / NextSeq / MiSeq), the MT worker set's kernel stats, then the default bench line itself.
#   tools/prof_r06.sh <tag>    -> gpurun_out/<tag>_*; condense here: for t in a indel hiseq nextseq miseq miseq-legacy; do
#                                  python tools/make_profile_summary.py gpurun_out/<tag>_$t <tag>_$t; done
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
tools/gpu_profile.sh ${TAG}_a > gpurun_out/${TAG}_a.log 2>&1
tools/clock_probe_pmc.sh ${TAG} > gpurun_out/${TAG}_clock.txt 2>&1
tools/prof_model.sh ${TAG}_indel "--indel 0.001 0.003" > gpurun_out/${TAG}_indel.log 2>&1
for m in hiseq nextseq miseq miseq-legacy; do
  tools/prof_model.sh ${TAG}_$m "--model $m" 6 > gpurun_out/${TAG}_$m.log 2>&1
done
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_mtset/stats -o stats --output-format csv -- python $GRAFT_REPO_ROOT/tools/mt_workers_speed.py novaseq 64 256 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_mtset.log 2>&1)
(time python bench.py) > gpurun_out/bench_${TAG}.txt 2> gpurun_out/bench_${TAG}.err
tail -n 1 gpurun_out/bench_${TAG}.txt; tail -4 gpurun_out/bench_${TAG}.err
