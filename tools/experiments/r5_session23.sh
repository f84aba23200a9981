#!/bin/bash
# round 5, session 23: small levels -- two levels per launch wherever valid (FVVDP_BAND_FUSE=1) against the default rule, 1080p and 4K
R=$(pwd); OUT=$R/gpurun_out/r5s23; mkdir -p $OUT
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'levels', g['levels_us_per_frame_median'], 'fin', g['finalize_us_per_frame'], 'all', g['us_per_frame_all_levels'], 'frac', g['hbm_frac_all_levels'], d['jod'])"; }
for rep in 1 2 3; do
  for f in default 1; do
    E=""; [ $f = 1 ] && E="FVVDP_BAND_FUSE=1"
    env $E timeout 300 python bench.py --width 1920 --height 1080 --display standard_fhd --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "1080p fuse=$f"
  done
done
for rep in 1 2; do
  for f in default 1; do
    E=""; [ $f = 1 ] && E="FVVDP_BAND_FUSE=1"
    env $E timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "4K fuse=$f"
  done
done
