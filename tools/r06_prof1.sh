#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in miseq hiseq; do
  ISS_MAIN_GROUP=0 tools/prof_model.sh r06x_${m}_legacy "--model $m" 6 > gpurun_out/r06x_${m}_legacy.log 2>&1
  tools/prof_model.sh r06x_${m}_g "--model $m" 6 > gpurun_out/r06x_${m}_g.log 2>&1
done
