#!/bin/bash
# round 5, session 15: are two slow level-0 candidates of one process of DIFFERENT classes?  pairwise half-buffer writes, 3 processes
R=$(pwd); OUT=$R/gpurun_out/r5s15; mkdir -p $OUT
for rep in 1 2 3; do
  timeout 280 $R/build_variants/k1_stream 4 4 0 2 > $OUT/stream$rep.txt 2>&1
  echo "== process $rep"; awk 'NF>20 && ($1 ~ /^[0-9]+$/) {print $1, $2, $4, $30}' $OUT/stream$rep.txt | head -12 | tr '\n' ';'; echo
  sed -n '/^pairwise/,/^checksum/p' $OUT/stream$rep.txt | grep -v checksum
done
