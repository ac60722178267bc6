cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== configs2"; bash tools/ab_multi.sh "" build_ab/libiss_new.so build_ab/libiss_top.so 2>&1 | tee gpurun_out/ab_b_main.log
echo "== indel"; bash tools/ab_multi.sh "--indel 0.001 0.003" build_ab/libiss_new.so build_ab/libiss_top.so 2>&1 | tee gpurun_out/ab_b_indel.log
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_indel_b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --indel 0.001 0.003 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads"
export ISS_SETUP_AHEAD=0
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $BENCH > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/sq1 -o p --output-format csv -- $BENCH > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_BRANCH --kernel-trace -d $OUT/sq2 -o p --output-format csv -- $BENCH > $OUT/sq2.log 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/sq*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in acc:
    if "iss::" in k: print(k, {c: "%.3g" % (acc[k][c]/n[k][c]) for c in sorted(acc[k])})
PY
head -8 $OUT/stats/stats_kernel_stats.csv | cut -c1-150
