#!/bin/bash
# round 5, session 20: level 0 in two chosen ranges (even / odd frame slots): state tests, the whole suite with every context split,
# then the bench pair with the choice on / off, alternating processes
R=$(pwd); OUT=$R/gpurun_out/r5s20; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_state.py -m gpu -x -q 2>&1 | tail -6
FVVDP_LEVEL0_SPLIT=1 timeout 1800 python -m pytest tests -m gpu -q -k "not level0_in_two and not bench" > $OUT/pytest_split.log 2>&1; tail -6 $OUT/pytest_split.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; a=d['level0_alloc']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'lv01', g['levels_us_per_frame_median'][0], 'all', g['us_per_frame_all_levels'], '|', a['in_use'][:110], a['pair_write_rate_tbs'], a['temporal_plus_pyramid_us_per_frame_at_creation'], 'first step', a['first_step_ms_incl_context_creation'], d['jod'])"; }
for rep in 1 2 3 4; do
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "two ranges"
  FVVDP_PLACEMENT_PROBE=0 timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "one range "
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
timeout 300 python tools/gpu_config4.py 2>/dev/null | grep "kernel us/frame:"
