#!/bin/bash
# round 5, session 32: the temporal kernel's loads and stores coupled in one wave against the same two streams decoupled (two kernels at once)
R=$(pwd); OUT=$R/gpurun_out/r5s32; mkdir -p $OUT
for rep in 1 2; do timeout 280 $R/build_variants/k1_stream 3 3 0 decoupled; echo; done > $OUT/decoupled.txt 2>&1
cat $OUT/decoupled.txt
