#!/bin/bash
# round 5, session 44: the tile kernel for small levels -- per-level times at 1080p and 4K for thresholds 0 (never), 40 k, 140 k pixels
R=$(pwd); OUT=$R/gpurun_out/r5s44; mkdir -p $OUT
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'levels', g['levels_us_per_frame_median'], 'fin', g['finalize_us_per_frame'], 'all', g['us_per_frame_all_levels'], 'frac', g['hbm_frac_all_levels'], d['jod'])"; }
for rep in 1 2 3; do
  for t in 0 40000 140000; do
    FVVDP_BAND_TILE=$t timeout 300 python bench.py --width 1920 --height 1080 --display standard_fhd --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "1080p tile<=$t"
  done
done
for rep in 1 2; do
  for t in 0 40000; do
    FVVDP_BAND_TILE=$t timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "4K tile<=$t"
  done
done
