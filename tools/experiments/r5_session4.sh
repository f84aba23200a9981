#!/bin/bash
# round 5, session 4: physical chunks of 1 GiB in allocation order: streaming-write rate alone and in pairs (which chunks share the
# resource that caps a single GiB at ~5.3 TB/s while a well-mixed 8 GB buffer reaches 7.0?)
R=$(pwd); OUT=$R/gpurun_out/r5s4; mkdir -p $OUT
$R/build_variants/k1_stream 200 zones > $OUT/zones.txt 2>&1
cat $OUT/zones.txt
