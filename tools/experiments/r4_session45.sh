#!/bin/bash
# round 4, session 45: several level-0 allocations held at once in ONE process: does the temporal kernel's speed differ between them?
R=$(pwd); OUT=$R/gpurun_out/r4s45; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0
( echo "# chunk-mapped"; python tools/experiments/gpu_alloc_draws.py 8 2>/dev/null
  echo "# hipMalloc"; FVVDP_ALLOC=malloc python tools/experiments/gpu_alloc_draws.py 8 2>/dev/null
  echo "# chunk-mapped, second process"; python tools/experiments/gpu_alloc_draws.py 8 2>/dev/null ) > $OUT/draws.txt
cat $OUT/draws.txt
