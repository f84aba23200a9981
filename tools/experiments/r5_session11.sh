#!/bin/bash
# round 5, session 11: fine-grained and uncached device memory as level-0 candidates next to hipMalloc / chunk mappings
R=$(pwd); OUT=$R/gpurun_out/r5s11; mkdir -p $OUT
$R/build_variants/k1_stream 3 3 0 3 > $OUT/stream.txt 2>&1
cut -c1-150 $OUT/stream.txt | head -20
awk '{print $1, $2, $4, $11, $30, $31}' $OUT/stream.txt | head -16
cd $R; python tools/experiments/gpu_predict_overhead.py 2>/dev/null | tee $OUT/predict_overhead.txt
