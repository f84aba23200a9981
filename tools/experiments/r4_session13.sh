#!/bin/bash
# round 4, session 13: foveated two-level pass (band2_fov_kernel): parity + A/B on configs[3]
R=$(pwd); OUT=$R/gpurun_out/r4s13; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k "foveated" > $OUT/pytest_fov2.txt 2>&1
tail -15 $OUT/pytest_fov2.txt
for rep in 1 2 3; do
  echo "== two-level" >> $OUT/fov_ab.txt
  timeout 300 python tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^Q_per_ch|^kernel us" | tail -3 >> $OUT/fov_ab.txt
  echo "== one level per launch (FVVDP_FOV_FUSE=0)" >> $OUT/fov_ab.txt
  FVVDP_FOV_FUSE=0 timeout 300 python tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^Q_per_ch|^kernel us" | tail -3 >> $OUT/fov_ab.txt
done
cat $OUT/fov_ab.txt
