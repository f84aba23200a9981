#!/bin/bash
# round 5, session 25: session 23's sequence again (1080p processes, then 4K ones) with the pair rates of every context's choice printed
R=$(pwd); OUT=$R/gpurun_out/r5s25; mkdir -p $OUT
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; a=d['level0_alloc']; k=d['roofline_k1']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'min/max', round(k['min_launch_ms']*1000/60,2), round(k['max_launch_ms']*1000/60,2), 'levels', g['levels_us_per_frame_median'], 'all', g['us_per_frame_all_levels'], '|', a['kept_indices'], a['pair_write_rate_tbs'], a['temporal_plus_pyramid_us_per_frame_at_creation'], d['jod'])"; }
for rep in 1 2 3; do
  for f in default 1; do
    E=""; [ $f = 1 ] && E="FVVDP_BAND_FUSE=1"
    env $E timeout 300 python bench.py --width 1920 --height 1080 --display standard_fhd --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "1080p fuse=$f"
  done
done
for rep in 1 2 3 4; do
  for f in default 1; do
    E=""; [ $f = 1 ] && E="FVVDP_BAND_FUSE=1"
    env $E timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "4K fuse=$f"
  done
done
