#!/bin/bash
# The same library with and without an environment switch, interleaved: tools/ab_env.sh "<bench args>" VAR=a VAR=b ...
ARGS=$1; shift
for rep in 1 2; do
  for E in "$@"; do
    env $E timeout 120 python bench.py $ARGS --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$E', 'value %.4g ms/step %.4f' % (d['value'], d['ms_per_step']), 'main %.4f' % k['main_ms'], 'parity', str(d.get('parity_window'))[:12])"
  done
done
