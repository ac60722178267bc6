#!/bin/bash
# PMC pass of the indel-heavy bench (BASELINE configs[4] flavour): what bounds k_indel_fixup?
OUT=$GRAFT_REPO_ROOT/gpurun_out/indel_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --indel 0.001 0.003 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/sq1 -o p --output-format csv -- $BENCH > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAVES --kernel-trace -d $OUT/sq2 -o p --output-format csv -- $BENCH > $OUT/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT 2>/dev/null | grep -A16 "k_indel_fixup"
grep -h "fixup\|k_indel_scan\|k_main" $OUT/stats/*kernel_stats.csv | cut -c1-140
