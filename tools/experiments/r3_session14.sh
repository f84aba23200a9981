#!/bin/bash
# foveated kernel: what a strip that owns 54 instead of 60 of its 64 lanes costs (timing only: results of strip54 are wrong on
# purpose -- neighbouring strips overlap), and the kernel without the rho map (fewer bytes, more VALU work)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s14
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in default strip54 norhomap; do
  unset FVVDP_LIB FVVDP_FOV_NO_RHOMAP
  [ $v = strip54 ] && export FVVDP_LIB=$R/build_variants/strip54.so
  [ $v = norhomap ] && export FVVDP_FOV_NO_RHOMAP=1
  echo "== $v ($rep)" >> $OUT/fov_lane_eff.txt
  timeout 200 python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^kernel" | tail -1 >> $OUT/fov_lane_eff.txt
done
done
unset FVVDP_LIB FVVDP_FOV_NO_RHOMAP
# the plain one-level kernel (memory-bound at level 0 in round 2) with the same change, for contrast
for v in default strip54; do
  unset FVVDP_LIB; [ $v = strip54 ] && export FVVDP_LIB=$R/build_variants/strip54.so
  echo "== plain one-level, $v" >> $OUT/fov_lane_eff.txt
  FVVDP_BAND_FUSE=0 timeout 200 python $R/tools/gpu_bandonly_speed.py 8 2>/dev/null | tail -2 >> $OUT/fov_lane_eff.txt
done
cat $OUT/fov_lane_eff.txt
