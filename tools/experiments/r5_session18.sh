#!/bin/bash
# round 5, session 18: does the foveated level-0 kernel (reads level 0, writes level 1) follow the class of LEVEL 1?
R=$(pwd); OUT=$R/gpurun_out/r5s18; mkdir -p $OUT
cd $R
for rep in 1 2 3 4 5 6; do
  FVVDP_DEBUG_VARIANT=1 timeout 300 python tools/gpu_config4.py 2>&1 | grep -E "kernel us/frame:|level 0 write rate|level 1 |level 2 " | cut -c1-160 | tr '\n' '|'; echo
done > $OUT/fov_l1.txt 2>&1
cat $OUT/fov_l1.txt
