#!/bin/bash
# Level 0 mapped twice (virtual-memory API): the readers (pyramid kernel) use a 1 GiB-aligned mapping (large page-table
# fragments), the temporal kernel writes through a second mapping of the SAME physical memory shifted by FVVDP_L0_ALIAS_MB
# (fragments capped at that size).  K1 / K2b per re-created context.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s19
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FVVDP_ALLOC=vmm FVVDP_VMM_ALIGN_MB=1024 FVVDP_DEBUG_ALLOC=1
for a in 0 2 64 4 512 0 2; do
  unset FVVDP_L0_ALIAS_MB; [ $a != 0 ] && export FVVDP_L0_ALIAS_MB=$a
  echo "== write alias shift $a MB" >> $OUT/alias.txt
  timeout 300 python $R/tools/experiments/gpu_k1_placement.py 2>&1 | grep -E "ctx|rror|alias" | head -6 >> $OUT/alias.txt
done
cat $OUT/alias.txt
