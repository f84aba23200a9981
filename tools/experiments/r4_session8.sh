#!/bin/bash
# round 4, session 8: buffer loads with scalar row offsets + v_fract/v_cvt_flr in the pyramid kernels: tests + same-box A/B
R=$(pwd); OUT=$R/gpurun_out/r4s8; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
for rep in 1 2 3; do
  echo "== new" >> $OUT/ab.txt
  timeout 300 python tools/gpu_bandonly_speed.py 12 2>/dev/null | grep -v amdgpu >> $OUT/ab.txt
  echo "== previous" >> $OUT/ab.txt
  FVVDP_LIB=$R/build_variants/r4_prev.so timeout 300 python tools/gpu_bandonly_speed.py 12 2>/dev/null | grep -v amdgpu >> $OUT/ab.txt
done
cat $OUT/ab.txt
timeout 300 python tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^Q_per_ch|^kernel us" > $OUT/fov_new.txt
FVVDP_LIB=$R/build_variants/r4_prev.so timeout 300 python tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^Q_per_ch|^kernel us" > $OUT/fov_prev.txt
echo "== fov new"; cat $OUT/fov_new.txt; echo "== fov prev"; cat $OUT/fov_prev.txt
