#!/bin/bash
# Guide bits 6 / 7 / 8 (ISS_GUIDE_BITS; the position tiles follow) for the shipped model families, k_main ms per 5 M pairs:
# the data behind the cost model of iss_model_upload (profiles/r04_ab_runs.txt).  Run on the GPU box.
cd $GRAFT_REPO_ROOT
run() { python bench.py $1 --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$2', 'value %.4g' % d['value'], 'main %.4f' % k['main_ms'])"; }
for m in hiseq nextseq miseq novaseq; do
  for rep in 1 2; do
  run "--model $m" "$m default"
  for gb in 6 7 8; do ISS_GUIDE_BITS=$gb run "--model $m" "$m GB=$gb"; done
  done
done
