#!/bin/bash
# round 5, session 37: among the best pairs by write rate, does the temporal kernel differ, and does timing it at creation pick a better pair?
R=$(pwd); OUT=$R/gpurun_out/r5s37; mkdir -p $OUT
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; a=d['level0_alloc']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'lv01', g['levels_us_per_frame_median'][0], '|', a['kept_indices'], a['pair_write_rate_tbs'], 'first step', a['first_step_ms_incl_context_creation'])"; }
for rep in 1 2 3 4 5 6 7 8; do
  FVVDP_PLACEMENT_K1PICK=4 FVVDP_DEBUG_VARIANT=1 timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2> $OUT/err.txt | line "pick of 4 "
  grep "level-0 pair" $OUT/err.txt | head -4 | sed 's/^/      /'
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "best rate "
done > $OUT/pick.txt 2>&1
cat $OUT/pick.txt
