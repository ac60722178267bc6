#!/bin/bash
# One bench leg under rocprofv3 on the GPU box: kernel-trace stats + the SQ / FETCH / WRITE counter passes (separate runs, as
# MI355X_MICROARCH.md prescribes), everything under gpurun_out/<tag>:   tools/prof_model.sh <tag> "<bench args>" [steps]
# Condense here afterwards: python tools/make_profile_summary.py gpurun_out/<tag> <tag>
TAG=${1:-prof}; ARGS=${2:-}; STEPS=${3:-8}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py $ARGS --steps $STEPS --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads"
echo "python bench.py $ARGS --steps $STEPS --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads" > $OUT/command.txt
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/sq1 -o p --output-format csv -- $BENCH > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES --kernel-trace -d $OUT/sq2 -o p --output-format csv -- $BENCH > $OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p --output-format csv -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p --output-format csv -- $BENCH > $OUT/write.log 2>&1
grep -h "k_main" $OUT/stats/*kernel_stats.csv | cut -c1-200 | head -3
