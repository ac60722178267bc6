#!/bin/bash
# guide bits (and the tiles that follow) under k_main_g: is round 4's cost model still right now that a deferred base costs less?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ab() { ARGS=$1; shift; for rep in 1 2; do for E in "$@"; do
    env $E timeout 120 python bench.py $ARGS --steps 30 --warmup 4 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$E', 'ms/step %.4f' % (d['ms_per_step']), 'main %.4f' % k['main_ms'], d['roofline']['kernel'], str(d.get('parity_window'))[:9])"
  done; done; }
for m in hiseq miseq nextseq; do echo "== $m"; ab "--model $m" ISS_X=0 ISS_GUIDE_BITS=6 ISS_GUIDE_BITS=7 ISS_GUIDE_BITS=8; done
echo "== miseq tiles"; ab "--model miseq" ISS_TILES=4 ISS_TILES=5 ISS_TILES=7 "ISS_TILES=10"
echo "== hiseq tiles"; ab "--model hiseq" "ISS_GUIDE_BITS=7 ISS_TILES=2" "ISS_GUIDE_BITS=8 ISS_TILES=2" "ISS_GUIDE_BITS=8 ISS_TILES=4" "ISS_GUIDE_BITS=8 ISS_TILES=4 ISS_MAIN_GROUP=4"
