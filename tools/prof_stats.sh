#!/bin/bash
# rocprofv3 kernel-trace stats only: tools/prof_stats.sh <tag> "<bench args>"
TAG=${1:-stats}; ARGS=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python $GRAFT_REPO_ROOT/bench.py $ARGS --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads > $OUT/stats.log 2>&1
head -9 $OUT/stats/stats_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
