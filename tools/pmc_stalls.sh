#!/bin/bash
# Where do k_main's wavefronts wait?  Stall / level counters of the SQ, TA, TCP (L1 incl. its TLB) and TCC (L2 -> memory) in separate
# passes (rocprofv3 --pmc), per k_main launch:   [PASSES="1 3 5"] tools/pmc_stalls.sh <tag> "<bench args>"   -> gpurun_out/<tag>/summary.txt
# (every pass under `timeout`: a counter set rocprofv3 cannot program aborts it and then hangs until the box's limit)
TAG=${1:-stalls}; ARGS=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py $ARGS --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads"
i=0
for SET in "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_IFETCH_LEVEL SQ_WAIT_ANY" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
           "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_WRITE_TAGCONFLICT_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_THRASHING_STALL TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_STALL_LFIFO_NO_RES" \
           "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_TAG_STALL TCC_IB_STALL TCC_BUSY TCC_EA0_WRREQ_LEVEL"; do
  i=$((i+1))
  [ -n "${PASSES:-}" ] && ! echo " $PASSES " | grep -q " $i " && continue
  timeout -k 5 150 rocprofv3 --pmc $SET --kernel-trace -d $OUT/p$i -o p --output-format csv -- $BENCH > $OUT/p$i.log 2>&1 || echo "pass $i: rc $? (a counter set rocprofv3 cannot take: see p$i.log)"
done
python3 - <<PY > $OUT/summary.txt
import csv,glob,collections
acc=collections.defaultdict(list); dur=[]
for f in sorted(glob.glob('$OUT/p*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'k_main' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] in ('SQ_WAVE_CYCLES','GRBM_GUI_ACTIVE'): dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
print('launch ms (profiled): %.4f' % (sum(dur)/max(len(dur),1)))
for k in sorted(acc): print('%-45s %d launches  %.4g per launch' % (k, len(acc[k]), sum(acc[k])/len(acc[k])))
PY
cat $OUT/summary.txt
