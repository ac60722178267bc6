#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_mt_compat.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fastq or cli or worker" 2>&1 | tail -3
echo "== temp files + concatenation (round 5)"; ISS_SET_TEMP_FILES=1 python tools/mt_set_e2e.py 64 16 2>&1 | tail -1
echo "== final files"; python tools/mt_set_e2e.py 64 16 2>&1 | tail -1
python tools/mt_set_e2e.py 64 64 2>&1 | tail -1
python tools/mt_set_e2e.py 256 64 2>&1 | tail -1
