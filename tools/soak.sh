#!/bin/bash
# Soak of the randomized GPU tests on other configurations than the suite's (ISS_FUZZ_OFFSET), everything captured:
#   tools/soak.sh <minutes> [first offset]   -> gpurun_out/soak/run_<offset>.log, failing runs keep their tmp dirs
MIN=${1:-4}; OFF=${2:-1000}
OUT=gpurun_out/soak; mkdir -p $OUT
END=$(( $(date +%s) + MIN * 60 )); RUNS=0; FAILS=0; EXECS=0
while [ $(date +%s) -lt $END ]; do
  LOG=$OUT/run_$OFF.log
  ISS_FUZZ_OFFSET=$OFF timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mt_compat.py -q -x --tb=long -p no:cacheprovider \
      -k "randomized or worker or gzip or tiny_jobs" --basetemp=$OUT/tmp_$OFF > $LOG 2>&1
  RC=$?
  N=$(grep -Eo "[0-9]+ passed" $LOG | grep -Eo "[0-9]+" | tail -1); EXECS=$(( EXECS + ${N:-0} ))
  if [ $RC -ne 0 ]; then FAILS=$(( FAILS + 1 )); echo "offset $OFF FAILED rc=$RC (log and files kept)"; else rm -rf $OUT/tmp_$OFF $LOG; fi
  RUNS=$(( RUNS + 1 )); OFF=$(( OFF + 1000 ))
done
echo "soak: $RUNS runs, $EXECS test executions, $FAILS failing runs" | tee $OUT/summary.txt
