#!/bin/bash
# bench.py on the shipped models and the indel-heavy synthetic one (kernel split + roofline fraction): tools/ab_models.sh [lib.so ...]
LIBS=${@:-insilicoseq_amd/libiss_mi355x.so}
run() {  # label, bench args
  for L in $LIBS; do
    ISS_MI355X_LIB=$PWD/$L timeout 200 python bench.py $2 --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', '$L'.split('/')[-1], 'value %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'], 'main %.3f scan %.3f setup %.3f fix %.3f' % (k['main_ms'], k['indel_scan_ms'] or 0, k['setup_ms'] or 0, k['indel_fixup_ms'] or 0), d['parity_window'][:2])"
  done
}
for m in novaseq hiseq miseq nextseq miseq-legacy; do run $m "--model $m"; done
run indel "--indel 0.001 0.003"
