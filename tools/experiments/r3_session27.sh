#!/bin/bash
# session 27: what bounds the pyramid kernels -- ablation builds (tools/build_variant.sh):
#   b2abl   -DBAND2_ABLATE=1     two-level kernel without the per-pixel tail (data flow only: loads, filters, stores)
#   b2nomem -DBAND2_ABLATE_MEM   two-level kernel with all arithmetic, rows re-read from 8 L2-resident rows, nothing stored
#   b2abl4  -DBAND2_ABLATE=1 -DBAND2_LB=4   data flow only at 4 waves per SIMD (15 spilled dwords)
#   abl1    -DBAND_ABLATE=1      one-level kernels (foveated and not) without the per-pixel tail
#   noecc2  -DFOV_ABLATE_ECC     foveated kernel without the eccentricity arithmetic (2 sqrt + ~10 VALU per pixel)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s27
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for L in default b2abl b2nomem b2abl4; do
    if [ $L = default ]; then unset FVVDP_LIB; else export FVVDP_LIB=$R/build_variants/$L.so; fi
    python $R/tools/gpu_bandonly_speed.py 12 2>/dev/null | grep -v Warn | tail -1 | sed "s/^.*bands us/bands us/" | cut -c1-120 | sed "s/^/$L two-level: /" | tee -a $OUT/band2_bounds.txt
  done
done
for rep in 1 2 3; do
  for L in default abl1; do
    if [ $L = default ]; then unset FVVDP_LIB; else export FVVDP_LIB=$R/build_variants/$L.so; fi
    FVVDP_BAND_FUSE=0 python $R/tools/gpu_bandonly_speed.py 12 2>/dev/null | grep -v Warn | tail -1 | sed "s/^.*bands us/bands us/" | cut -c1-120 | sed "s/^/$L one-level: /" | tee -a $OUT/band1_bounds.txt
  done
  for L in default abl1 noecc2; do
    if [ $L = default ]; then unset FVVDP_LIB; else export FVVDP_LIB=$R/build_variants/$L.so; fi
    python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^kernel us" | tail -1 | sed "s/^/$L foveated: /" | tee -a $OUT/fov_bounds.txt
  done
done
