#!/bin/bash
# guide bits x position tiles sweep (ISS_GUIDE_BITS / ISS_TILES tuning aids): tools/tile_sweep.sh model "gb:tiles gb:tiles ..."
m=$1; shift
for c in $@; do
  gb=${c%%:*}; t=${c##*:}
  ISS_GUIDE_BITS=$gb ISS_TILES=$t ISS_DEBUG_MODEL=1 timeout 120 python bench.py --model $m --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads 2> /tmp/sweep.err | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$m gb $gb tiles $t', 'value %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'], 'main %.3f' % k['main_ms'])"
  grep "^\[model\] RL" /tmp/sweep.err | head -1
done
