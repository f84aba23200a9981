#!/bin/bash
# round 5, session 39: chunk heights of the two-level pyramid kernel on the final build (tickets on): rows of level C per chunk, rows of the short chunks
R=$(pwd); OUT=$R/gpurun_out/r5s39; mkdir -p $OUT
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; r=d['roofline']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'lv01 median', g['levels_us_per_frame_median'][0], 'min', round(r['min_launch_ms']*1000/60,2), 'all', g['us_per_frame_all_levels'])"; }
for rep in 1 2; do
for kr in default 20 24 27 30 34 39 45 54 68; do
  E=""; [ $kr != default ] && E="FVVDP_BAND2_KR=$kr"
  env $E timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "kr=$kr"
done
done > $OUT/kr.txt 2>&1
cat $OUT/kr.txt
