#!/bin/bash
# round 4, session 23: the foveated one-level kernel with its 4 waves in step on adjacent strips (FOV_LOCKSTEP builds)
R=$(pwd); OUT=$R/gpurun_out/r4s23; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0
FVVDP_LIB=$R/build_variants/fovls1.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu -k "fov" > $OUT/tests_fovls1.txt 2>&1
tail -n 3 $OUT/tests_fovls1.txt
rm -f $OUT/fov.txt
for rep in 1 2; do
  for v in base fovls1 fovls2; do
    echo "== $v" >> $OUT/fov.txt
    FVVDP_LIB=$R/build_variants/$v.so python tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^kernel us" >> $OUT/fov.txt
  done
done
cat $OUT/fov.txt
