#!/bin/bash
# round 5, session 12: uncached device memory as level 0 -- K1 and the pyramid pass per kind (no choice: FVVDP_PLACEMENT_PROBE=0 for
# chunk-mapped / hipMalloc; FVVDP_ALLOC=uncached forces the kind), alternating processes; then the default (best of 4 kinds)
R=$(pwd); OUT=$R/gpurun_out/r5s12; mkdir -p $OUT
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; print('$1', d['ms_per_step'], 'predict', d['predict_call_ms'], 'K1', g['temporal_us_per_frame_median'], 'levels', g['levels_us_per_frame_median'][:3], 'all', g['us_per_frame_all_levels'], d['level0_alloc']['in_use'][:12], d['level0_alloc']['candidates_us_per_frame'], d['jod'])"; }
for rep in 1 2 3; do
  FVVDP_PLACEMENT_PROBE=0 timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "chunks  "
  FVVDP_PLACEMENT_PROBE=0 FVVDP_ALLOC=malloc timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "malloc  "
  FVVDP_PLACEMENT_PROBE=1 FVVDP_ALLOC=uncached timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "uncached"
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "best-of4"
done > $OUT/kinds.txt 2>&1
cat $OUT/kinds.txt
FVVDP_ALLOC=uncached timeout 300 python tools/gpu_config4.py 2>/dev/null | grep "kernel us/frame:" | sed "s/^/uncached fov /"
timeout 300 python tools/gpu_config4.py 2>/dev/null | grep "kernel us/frame:" | sed "s/^/default fov  /"
timeout 600 python -m pytest tests/test_gpu_state.py -m gpu -x -q 2>&1 | tail -3
