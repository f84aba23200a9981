#!/bin/bash
# session 29: YUV ingest with the source loads in a fixed order (counted vmcnt waits instead of a drain behind the stores)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s29
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for L in default "$@"; do
  if [ $L = default ]; then unset FVVDP_LIB; else export FVVDP_LIB=$R/build_variants/$L.so; fi
  python $R/tools/gpu_yuv.py 2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444 2160x3840x60:10:444:60 2160x3840x60:10:420 2>/dev/null | grep temporal | sed "s/^libfvvdp_hip.so/$L:/" | tee -a $OUT/yuv_ab.txt
done
done
