#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ab() { ARGS=$1; shift; for rep in 1 2 3; do for E in "$@"; do
    env $E timeout 120 python bench.py $ARGS --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$E', 'ms/step %.4f' % (d['ms_per_step']), 'main %.4f' % k['main_ms'], str(d.get('parity_window'))[:9])"
  done; done; }
# ISS_ABL bits: 1 drop held patches, 2 no rounds at all, 4 rows leave at once
for m in "" "--model hiseq"; do
  echo "== $m"
  ab "$m" ISS_MAIN_GROUP=0 ISS_ABL=0 ISS_ABL=1 ISS_ABL=2 ISS_ABL=4 ISS_ABL=5 ISS_ABL=6
done
