#!/bin/bash
# round 5, session 22: the whole GPU suite on the two-range build, then ten pairs of bench processes, level 0 in two chosen ranges / in one
R=$(pwd); OUT=$R/gpurun_out/r5s22; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; a=d['level0_alloc']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'lv01', g['levels_us_per_frame_median'][0], 'all', g['us_per_frame_all_levels'], '|', a['kept_indices'], a['pair_write_rate_tbs'], a['temporal_plus_pyramid_us_per_frame_at_creation'], 'first step', a['first_step_ms_incl_context_creation'], d['jod'])"; }
for rep in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "two ranges"
  FVVDP_PLACEMENT_PROBE=0 timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "one range "
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
