#!/bin/bash
# K1 / K2b against the allocation flags of the pyramid scratch (hipExtMallocWithFlags): is the placement mode a caching attribute?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in default contiguous finegrained uncached default; do
  unset FVVDP_ALLOC_FLAGS; [ $f != default ] && export FVVDP_ALLOC_FLAGS=$f
  echo "== FVVDP_ALLOC_FLAGS=$f" >> $OUT/alloc_flags.txt
  timeout 300 python $R/tools/experiments/gpu_k1_placement.py 2>&1 | grep -E "ctx|rror" >> $OUT/alloc_flags.txt
done
cat $OUT/alloc_flags.txt
