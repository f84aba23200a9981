#!/bin/bash
# round 4, session 7: floors of the two-level kernel after the clamp-free variant (ablation builds, chunk-mapped scratch)
R=$(pwd); OUT=$R/gpurun_out/r4s7; mkdir -p $OUT
cd $R
for rep in 1 2; do
for v in default r4_ablate_tail r4_ablate_mem r4_lb4; do
  for inr in 1 0; do
    if [ $v = default ]; then L=""; else L="FVVDP_LIB=$R/build_variants/$v.so"; fi
    echo "== $v inrange=$inr" >> $OUT/floors.txt
    env $L FVVDP_BAND_INRANGE=$inr timeout 300 python tools/gpu_bandonly_speed.py 10 2>/dev/null | grep -v amdgpu >> $OUT/floors.txt
  done
done
done
cat $OUT/floors.txt
