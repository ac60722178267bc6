#!/bin/bash
# the chip's clock under k_main: GRBM_GUI_ACTIVE (cycles the GPU is active) per launch / the launch's duration
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06}_clock; mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads"
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d $OUT/grbm -o p --output-format csv -- $BENCH > $OUT/grbm.log 2>&1
ls $OUT/grbm; head -3 $OUT/grbm/*counter_collection.csv
python3 - <<PY
import csv,glob,collections
f=glob.glob('$OUT/grbm/*counter_collection.csv')[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_main' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
(rocm-smi --showclocks --showpower 2>&1 | head -30) || true
