#!/bin/bash
# round 5, session 6: as session 5 with physical chunks of 1 GiB (session 5: 120 GiB of 32 MB chunks all came from ONE class), then
# 256 MB and 64 MB chunks: at which handle size does the allocator hand out both classes?
R=$(pwd); OUT=$R/gpurun_out/r5s6; mkdir -p $OUT
for MB in 1024 256 64 1024; do
  $R/build_variants/k1_stream 100 $MB balanced > $OUT/balanced_$MB.txt 2>&1
  cat $OUT/balanced_$MB.txt
done
