#!/bin/bash
# round 5, session 36: the final build on one more box -- six default bench processes and two with one range as allocated
R=$(pwd); OUT=$R/gpurun_out/r5s36; mkdir -p $OUT
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; a=d['level0_alloc']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'lv01', g['levels_us_per_frame_median'][0], 'all', g['us_per_frame_all_levels'], '|', a['kept_indices'], a.get('further_candidates_tried'), a['pair_write_rate_tbs'], d['jod'])"; }
for rep in 1 2 3 4 5 6; do timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "two ranges"; done
for rep in 1 2; do FVVDP_PLACEMENT_PROBE=0 timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "one range "; done
