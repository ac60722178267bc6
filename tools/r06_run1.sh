#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for m in novaseq hiseq miseq nextseq; do ISS_DEBUG_MODEL=1 timeout 100 python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads 2>&1 | grep "^\[model\] per base" | head -1; done
timeout 600 python bench.py > gpurun_out/r06_bench1.json 2> gpurun_out/r06_bench1.err; tail -c 3000 gpurun_out/r06_bench1.json
