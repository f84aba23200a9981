#!/bin/bash
# round 5, session 16: the level 0 across both classes of physical memory (balance_level0) -- six bench processes in a row with the
# search on, two with it off; then the state tests
R=$(pwd); OUT=$R/gpurun_out/r5s16; mkdir -p $OUT
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; a=d['level0_alloc']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'lv01', g['levels_us_per_frame_median'][0], 'all', g['us_per_frame_all_levels'], a['in_use'][:40], a['write_rate_tbs'], a['candidates_us_per_frame'], 'first step', a['first_step_ms_incl_context_creation'], d['jod'])"; }
for rep in 1 2 3; do
  FVVDP_DEBUG_VARIANT=1 timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2> $OUT/err$rep.txt | line "balance on "
  grep "level 0 write rate" $OUT/err$rep.txt | head -2
  FVVDP_LEVEL0_BALANCE=0 timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "balance off"
done > $OUT/balance.txt 2>&1
cat $OUT/balance.txt
timeout 300 python tools/gpu_config4.py 2>/dev/null | grep "kernel us/frame:"
timeout 900 python -m pytest tests/test_gpu_state.py -m gpu -x -q 2>&1 | tail -5
