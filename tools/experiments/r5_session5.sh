#!/bin/bash
# round 5, session 5: level-0 candidates assembled from physical chunks of known class (session 4: two classes of physical memory,
# ~5.4 TB/s of streaming writes into one class, 7.0 into both at once): which layouts put the temporal kernel in its fast mode?
R=$(pwd); OUT=$R/gpurun_out/r5s5; mkdir -p $OUT
$R/build_variants/k1_stream 120 balanced > $OUT/balanced.txt 2>&1
cat $OUT/balanced.txt
$R/build_variants/k1_stream 120 balanced > $OUT/balanced2.txt 2>&1
cat $OUT/balanced2.txt
