#!/bin/bash
# K1 / K2b against a padded level-0 slot stride (FVVDP_L0_SLOT_PAD_KB) on physically contiguous scratch (K1's slow mode):
# is the slow mode a resonance of the 132,710,400-byte frame stride?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s18
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && FVVDP_L0_SLOT_PAD_KB=68 timeout 900 python -m pytest tests -m gpu -x -q -k "parity or fused or sizes or cabi" > $OUT/pytest_pad68.log 2>&1; echo "rc $?" >> $OUT/pytest_pad68.log )
export FVVDP_ALLOC_FLAGS=contiguous
for pad in 0 4 64 256 1024 1472 2048 8192 65536 16 1000; do
  export FVVDP_L0_SLOT_PAD_KB=$pad
  echo "== contiguous, slot pad $pad KB" >> $OUT/slot_pad.txt
  timeout 300 python $R/tools/experiments/gpu_k1_placement.py 2>&1 | grep -E "ctx|rror" | head -3 >> $OUT/slot_pad.txt
done
tail -3 $OUT/pytest_pad68.log; cat $OUT/slot_pad.txt
