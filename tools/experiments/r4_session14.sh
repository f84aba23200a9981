#!/bin/bash
# round 4, session 14: variants of the foveated two-level pass on configs[3] (4K x120, moving gaze, PQ)
R=$(pwd); OUT=$R/gpurun_out/r4s14; mkdir -p $OUT
cd $R
for rep in 1 2; do
  for v in default r4_fov_p1 r4_fov_p1w6 r4_fov_p2w6; do
    if [ $v = default ]; then L=""; else L="FVVDP_LIB=$R/build_variants/$v.so"; fi
    echo "== $v (two-level)" >> $OUT/fov_variants.txt
    env $L timeout 300 python tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^kernel us" | tail -2 >> $OUT/fov_variants.txt
  done
  echo "== one level per launch (FVVDP_FOV_FUSE=0)" >> $OUT/fov_variants.txt
  FVVDP_FOV_FUSE=0 timeout 300 python tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^kernel us" | tail -2 >> $OUT/fov_variants.txt
done
cat $OUT/fov_variants.txt
