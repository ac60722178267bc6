#!/bin/bash
# A/B of library builds on the GPU box, interleaved, three passes: tools/ab_libs.sh "<bench args>" lib1.so lib2.so ...  (k_main ms per step from HIP events)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for rep in 1 2 3; do for L in "$@"; do
  ISS_MI355X_LIB=$PWD/$L timeout 120 python bench.py $ARGS --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$L', 'ms/step %.4f' % (d['ms_per_step']), 'main %.4f' % k['main_ms'], str(d.get('parity_window'))[:9])"
done; done
