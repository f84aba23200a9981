#!/bin/bash
# round 5, session 28: further candidates where the first six lie in one class -- state tests, then eight contexts in one process (three processes)
R=$(pwd); OUT=$R/gpurun_out/r5s28; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_state.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2 3; do
  echo "== two chosen ranges (default)"; FVVDP_DEBUG_VARIANT=1 timeout 600 python tools/experiments/gpu_alloc_draws.py 8 2>&1 | grep -E "round|level 0"
done > $OUT/draws.txt 2>&1
cat $OUT/draws.txt
