#!/bin/bash
# One short bench run per library build: tools/ab_once.sh "<bench args>" lib1.so lib2.so ...   (kernel split per step)
ARGS=$1; shift
for L in "$@"; do
  ISS_MI355X_LIB=$PWD/$L timeout 120 python bench.py $ARGS --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$L', 'value %.4g' % d['value'], 'main %.4f scan %.4f setup %.4f fix %.4f' % (k['main_ms'], k['indel_scan_ms'] or 0, k['setup_ms'] or 0, k['indel_fixup_ms'] or 0), 'parity', str(d.get('parity_window'))[:40])"
done
