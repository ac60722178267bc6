#!/bin/bash
# bench.py on the other shipped models (kernel split + roofline fraction): tools/other_models.sh [models...]
for m in ${@:-hiseq miseq nextseq miseq-legacy}; do
  ISS_DEBUG_MODEL=1 timeout 200 python bench.py --model $m --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads 2> gpurun_out/model_$m.err | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$m', 'value %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'], 'main %.3f scan %.3f setup %.3f fix %.3f' % (k['main_ms'], k['indel_scan_ms'] or 0, k['setup_ms'] or 0, k['indel_fixup_ms'] or 0), d['parity_window'][:2])"
  grep "^\[model\]" gpurun_out/model_$m.err | head -2
done
