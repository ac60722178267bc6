#!/bin/bash
# Round-4 profile set on the GPU box (via gpurun): default bench (stats, PMC, traffic + calibration), the indel-heavy bench
# (stats, SQ counters, traffic), rocprofv3 stats of the other shipped models, then the default bench line itself.
#   tools/prof_r04.sh <tag>      -> gpurun_out/<tag>_*; condense with tools/make_profile_summary.py here afterwards
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
tools/gpu_profile.sh ${TAG}_a > gpurun_out/${TAG}_a.log 2>&1
tools/prof_indel.sh ${TAG}_indel > gpurun_out/${TAG}_indel.log 2>&1
tools/pmc_mem.sh ${TAG}_indel_mem "--indel 0.001 0.003" > gpurun_out/${TAG}_indel_mem.log 2>&1
for m in hiseq nextseq miseq miseq-legacy; do
  tools/prof_stats.sh ${TAG}_$m "--model $m" > gpurun_out/${TAG}_$m.log 2>&1
  echo "python bench.py --model $m --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads" > gpurun_out/${TAG}_$m/command.txt
done
(time python bench.py) > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
tail -c 600 gpurun_out/bench_${TAG}.json; tail -4 gpurun_out/bench_${TAG}.err
