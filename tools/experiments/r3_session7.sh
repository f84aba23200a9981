#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s7
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
OFFS=0,4,8,12,16,20,24,28,32,36,40,44,48,52,56,60,64,1,2,3,5,6,7,0 timeout 900 python $R/tools/experiments/gpu_k1_offset_sweep.py 6 > $OUT/k1_fine.txt 2>&1
ls -la $OUT
