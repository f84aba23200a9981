#!/bin/bash
# round 5, session 19: levels 0-2 across both classes (balance_levels): configs[3] and the bench pair, with the search on / off, and the state tests
R=$(pwd); OUT=$R/gpurun_out/r5s19; mkdir -p $OUT
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; a=d['level0_alloc']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'levels', g['levels_us_per_frame_median'][:3], 'all', g['us_per_frame_all_levels'], a['in_use'][:30], a['write_rate_tbs'], a['candidates_us_per_frame'], 'first step', a['first_step_ms_incl_context_creation'], d['jod'])"; }
for rep in 1 2 3 4; do
  FVVDP_DEBUG_VARIANT=1 timeout 300 python tools/gpu_config4.py 2>&1 | grep -E "kernel us/frame:|level write rates" | cut -c1-260 | sed "s/^/fov on  /"
  FVVDP_LEVEL0_BALANCE=0 timeout 300 python tools/gpu_config4.py 2>&1 | grep -E "kernel us/frame:" | sed "s/^/fov off /"
  FVVDP_DEBUG_VARIANT=1 timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2> $OUT/err.txt | line "4K on  "
  grep "level write rates" $OUT/err.txt | head -1 | cut -c1-260
  FVVDP_LEVEL0_BALANCE=0 timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "4K off "
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
timeout 900 python -m pytest tests/test_gpu_state.py -m gpu -x -q 2>&1 | tail -5
