#!/bin/bash
# round 5, session 27: VERDICT r4 item 1's own bar -- eight contexts held in ONE process, K1 per context: level 0 in two chosen ranges
# (default) against one range as allocated (chunk-mapped; hipMalloc), twice each
R=$(pwd); OUT=$R/gpurun_out/r5s27; mkdir -p $OUT
cd $R
for rep in 1 2; do
  echo "== two chosen ranges (default)"; timeout 600 python tools/experiments/gpu_alloc_draws.py 8 2>&1 | grep -E "round|level 0"
  echo "== one range, chunk-mapped"; FVVDP_PLACEMENT_PROBE=0 timeout 600 python tools/experiments/gpu_alloc_draws.py 8 2>&1 | grep -E "round|level 0"
  echo "== one range, hipMalloc"; FVVDP_PLACEMENT_PROBE=0 FVVDP_ALLOC=malloc timeout 600 python tools/experiments/gpu_alloc_draws.py 8 2>&1 | grep -E "round|level 0"
done > $OUT/draws.txt 2>&1
cat $OUT/draws.txt
