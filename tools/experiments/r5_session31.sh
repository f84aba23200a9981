#!/bin/bash
# round 5, session 31: does a read stream pay less for a write stream mixed into it when the two ranges lie in different classes of memory?
R=$(pwd); OUT=$R/gpurun_out/r5s31; mkdir -p $OUT
for rep in 1 2; do timeout 250 $R/build_variants/rw_classes 8; echo; done > $OUT/rw.txt 2>&1
cat $OUT/rw.txt
