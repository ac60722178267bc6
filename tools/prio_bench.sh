#!/bin/bash
# the default bench (short) several times per setup-stream priority: how often does a process land on the slow level?
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6; do
  for P in -1 0 1; do
    ISS_SETUP_PRIO=$P timeout 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('prio $P rep $rep ms/step %.4f main %.4f' % (d['ms_per_step'], k['main_ms']))"
  done
done | sort -k2,2n -s
