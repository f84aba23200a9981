#!/bin/bash
# session 25: repeated A/B of foveated-kernel variants (fresh process = fresh allocation each time)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s25
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3 4 5; do
for L in default "$@"; do
  if [ $L = default ]; then unset FVVDP_LIB; else export FVVDP_LIB=$R/build_variants/$L.so; fi
  python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^kernel us" | tail -1 | sed "s/^/$L: /" | tee -a $OUT/ab.txt
done
done
