#!/bin/bash
# round 5, session 17: configs[3] (foveated one-level kernels read level 0) on a balanced level 0 against the other choices
R=$(pwd); OUT=$R/gpurun_out/r5s17; mkdir -p $OUT
cd $R
for rep in 1 2 3; do
  FVVDP_DEBUG_VARIANT=1 timeout 300 python tools/gpu_config4.py 2>&1 | grep -E "kernel us/frame:|level 0 write rate|kept" | sed "s/^/balance on  /" | cut -c1-220
  FVVDP_LEVEL0_BALANCE=0 FVVDP_DEBUG_VARIANT=1 timeout 300 python tools/gpu_config4.py 2>&1 | grep -E "kernel us/frame:|kept" | sed "s/^/best of six /" | cut -c1-220
  FVVDP_PLACEMENT_PROBE=0 timeout 300 python tools/gpu_config4.py 2>&1 | grep -E "kernel us/frame:" | sed "s/^/plain       /"
done > $OUT/fov.txt 2>&1
cat $OUT/fov.txt
