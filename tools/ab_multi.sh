#!/bin/bash
# A/B of several builds of the library on the GPU box: tools/ab_multi.sh "<bench args>" lib1.so lib2.so ...; interleaved runs.
ARGS=$1; shift
for rep in 1 2; do
  for L in "$@"; do
    ISS_MI355X_LIB=$PWD/$L timeout 120 python bench.py $ARGS --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>&1 | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$L', 'value %.4g' % d['value'], 'main %.4f scan %.4f setup %.4f fix %.4f' % (k['main_ms'], k['indel_scan_ms'] or 0, k['setup_ms'] or 0, k['indel_fixup_ms'] or 0))"
  done
done
