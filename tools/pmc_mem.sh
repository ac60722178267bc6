#!/bin/bash
# HBM-side traffic per kernel (separate FETCH_SIZE / WRITE_SIZE passes): tools/pmc_mem.sh <tag> "<bench args>"
TAG=${1:-mem}; ARGS=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py $ARGS --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p --output-format csv -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p --output-format csv -- $BENCH > $OUT/write.log 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in acc:
    if "iss::" in k: print(k, {c: "%.4g" % (acc[k][c]/n[k][c]) for c in sorted(acc[k])})
PY
