#!/bin/bash
# round 4, session 19: occupancy over one launch of band2_kernel (profiling build with per-workgroup clocks)
R=$(pwd); OUT=$R/gpurun_out/r4s19; mkdir -p $OUT
cd $R
FVVDP_LIB=$R/build_variants/timeline.so python tools/gpu_timeline.py $OUT/timeline_4k.npy > $OUT/timeline_4k.txt 2>$OUT/err.txt
cat $OUT/timeline_4k.txt
tail -n 3 $OUT/err.txt
HH=1080 WW=1920 FVVDP_LIB=$R/build_variants/timeline.so python tools/gpu_timeline.py > $OUT/timeline_fhd.txt 2>>$OUT/err.txt
cat $OUT/timeline_fhd.txt
