#!/bin/bash
# round 5, session 7: level-0 candidates whose two halves are allocated S GiB apart (spacer allocated in between, then released) and
# interleaved chunk by chunk: does some S always reach the fast mode, whatever state the allocator is in?  Several processes in a row,
# three chunk sizes, and once more after a process that allocated and freed 150 GiB.
R=$(pwd); OUT=$R/gpurun_out/r5s7; mkdir -p $OUT
B=$R/build_variants/k1_stream
for MB in 32 256 32 1024 64; do echo "== chunk $MB MB"; $B $MB spread; done > $OUT/spread.txt 2>&1
$B 150 zones > /dev/null 2>&1
for MB in 32 256; do echo "== after 150 GiB churn, chunk $MB MB"; $B $MB spread; done >> $OUT/spread.txt 2>&1
cat $OUT/spread.txt
