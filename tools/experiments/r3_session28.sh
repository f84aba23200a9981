#!/bin/bash
# session 28: recorded A/B of the foveated kernel: prev (commit 005ba66: shifted window, map loads behind the row prefetch, strip-fastest
# order) / nofirst (HEAD with -DFOV_FRAME_FASTEST=0) / default (HEAD)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s28
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3 4 5; do
for L in prev nofirst default; do
  if [ $L = default ]; then unset FVVDP_LIB; else export FVVDP_LIB=$R/build_variants/$L.so; fi
  python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^kernel us" | tail -2 | tr '\n' ' ' | sed "s/^/$L: /; s/config4 4Kx120 foveated PQ: //; s/JOD.*kernel/| kernel/" | tee -a $OUT/ab.txt
  echo | tee -a $OUT/ab.txt
done
done
