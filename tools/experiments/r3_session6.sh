#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/tools/experiments/gpu_k1_probe_corr.py 14 > $OUT/k1_probe.txt 2>&1
( cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log )
ls -la $OUT
