#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_mt_compat.py -x -q -m gpu -k "configs0 or basic" 2>&1 | tail -3
echo "== configs3 on one GPU: chunk sizes"
for rep in 1 2; do for E in "ISS_MAIN_GROUP=0" "ISS_MAIN_GROUP=0 ISS_CHUNK_PAIRS=5000000" "ISS_MAIN_GROUP=0 ISS_CHUNK_PAIRS=12500000" "ISS_X=1" "ISS_CHUNK_PAIRS=5000000" "ISS_CHUNK_PAIRS=12500000"; do
  env $E timeout 300 python bench.py --workload configs3 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$E', 'value %.4g ms/step %.3f' % (d['value'], d['ms_per_step']), 'main %.3f setup %.3f' % (k['main_ms'], k['setup_ms'] or 0), d['roofline']['kernel'], str(d.get('parity_window'))[:9])"
done; done
echo "== default bench"
timeout 900 python bench.py > gpurun_out/r06_bench2.txt 2> gpurun_out/r06_bench2.err; tail -n 1 gpurun_out/r06_bench2.txt | wc -c; tail -n 1 gpurun_out/r06_bench2.txt
