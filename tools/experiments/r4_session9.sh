#!/bin/bash
# round 4, session 9: K1 on the 120-frame context (configs[3]): chunk sizes
R=$(pwd); OUT=$R/gpurun_out/r4s9; mkdir -p $OUT
cd $R
for mb in 2 8 32 128; do
  echo "== chunk $mb MB" >> $OUT/fov_chunks.txt
  FVVDP_VMM_CHUNK_MB=$mb timeout 300 python tools/gpu_config4.py 2>/dev/null | grep -E "^kernel us" >> $OUT/fov_chunks.txt
done
echo "== malloc" >> $OUT/fov_chunks.txt
FVVDP_ALLOC=malloc timeout 300 python tools/gpu_config4.py 2>/dev/null | grep -E "^kernel us" >> $OUT/fov_chunks.txt
cat $OUT/fov_chunks.txt
python tools/gpu_fps.py 30:60:u8 30:120:u8 60:120:u8 120:120:u8 2>/dev/null | grep -v Warn
FVVDP_ALLOC=malloc python tools/gpu_fps.py 30:60:u8 30:120:u8 60:120:u8 120:120:u8 2>/dev/null | grep -v Warn
