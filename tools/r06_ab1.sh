#!/bin/bash
# round 6: k_main_g (rows of a group wait in registers, patches in time) against k_main, interleaved on one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
for m in "" "--model hiseq" "--model miseq" "--model nextseq"; do
  echo "== $m"
  tools/ab_env.sh "$m" ISS_MAIN_GROUP=0 ISS_MAIN_GROUP_MIN=1 ISS_MAIN_GROUP_MIN=16 ISS_MAIN_GROUP_MIN=32 ISS_MAIN_GROUP_MIN=64
done
echo "== hiseq NP=1"
tools/ab_env.sh "--model hiseq" "ISS_MAIN_GROUP=1 ISS_MAIN_GROUP_MIN=1" "ISS_MAIN_GROUP=1 ISS_MAIN_GROUP_MIN=32"
