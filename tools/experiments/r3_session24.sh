#!/bin/bash
# session 24: band_item with the row window as an 8-slot ring (4 unrolled loop bodies) vs the shifted window
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s24
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
V=${1:-ring8}
for rep in 1 2; do
for L in default $V; do
  if [ $L = default ]; then unset FVVDP_LIB; else export FVVDP_LIB=$R/build_variants/$L.so; fi
  python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^kernel us" | tail -2 | sed "s/^/$L: /" | tee -a $OUT/ring8.txt
  FVVDP_BAND_FUSE=0 python $R/tools/gpu_bandonly_speed.py 12 2>/dev/null | grep -v Warn | tail -1 | sed "s/^/$L onelevel: /" | cut -c1-200 | tee -a $OUT/ring8.txt
done
done
export FVVDP_LIB=$R/build_variants/$V.so
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
