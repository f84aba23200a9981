#!/bin/bash
# K1 / K2b against the physical layout of the pyramid scratch, built with the virtual-memory API: chunk size and k-way
# interleave of the physical chunks inside the virtual range (contiguous allocations are always in K1's slow mode)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s17
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FVVDP_ALLOC=vmm
for cfg in "0:1" "2:1" "2:2" "2:8" "64:1" "64:2" "64:8" "1024:2" "16:4" "256:4"; do
  ch=${cfg%%:*}; k=${cfg##*:}
  unset FVVDP_VMM_CHUNK_MB FVVDP_VMM_INTERLEAVE
  [ $ch != 0 ] && export FVVDP_VMM_CHUNK_MB=$ch
  export FVVDP_VMM_INTERLEAVE=$k
  echo "== chunk ${ch} MB, interleave $k" >> $OUT/vmm_layout.txt
  timeout 400 python $R/tools/experiments/gpu_k1_placement.py 2>&1 | grep -E "ctx|rror" | head -5 >> $OUT/vmm_layout.txt
done
cat $OUT/vmm_layout.txt
