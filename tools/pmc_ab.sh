#!/bin/bash
# PMC comparison of library builds: tools/pmc_ab.sh tag lib1.so lib2.so ...  (k_main counters, 10 steps each)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  N=$(basename $L .so)
  OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/$N
  mkdir -p $OUT
  BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads"
  ISS_MI355X_LIB=$GRAFT_REPO_ROOT/$L rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/sq1 -o p --output-format csv -- $BENCH > $OUT/sq1.log 2>&1
  ISS_MI355X_LIB=$GRAFT_REPO_ROOT/$L rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH --kernel-trace -d $OUT/sq2 -o p --output-format csv -- $BENCH > $OUT/sq2.log 2>&1
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_main" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print("$N", {k: "%.4g" % (acc[k]/n[k]) for k in sorted(acc)})
PY
done
