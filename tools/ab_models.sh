#!/bin/bash
# tools/ab_models.sh a.so b.so: interleaved runs over the shipped models
for m in novaseq nextseq hiseq miseq; do for L in "$@"; do
ISS_MI355X_LIB=$PWD/$L python bench.py --model $m --steps 15 --warmup 3 --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$m $L', 'value %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'], 'main %.4f' % k['main_ms'])"
done; done
