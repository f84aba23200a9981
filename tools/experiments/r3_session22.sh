#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s22
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_max_sizes.py -x -q --durations=5 > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt
