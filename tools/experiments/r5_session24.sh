#!/bin/bash
# round 5, session 24: twelve 4K bench processes on the two-range build: K1 per process next to the pair rates of the choice
R=$(pwd); OUT=$R/gpurun_out/r5s24; mkdir -p $OUT
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; a=d['level0_alloc']; k=d['roofline_k1']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], [round(v*1000/60,1) for v in k['launch_ms_all']], 'lv01', g['levels_us_per_frame_median'][0], '|', a['kept_indices'], a['pair_write_rate_tbs'], a['temporal_plus_pyramid_us_per_frame_at_creation'], d['jod'])"; }
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "two ranges"
done > $OUT/k1.txt 2>&1
cat $OUT/k1.txt
