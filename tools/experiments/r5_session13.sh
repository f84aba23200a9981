#!/bin/bash
# round 5, session 13: how often is an uncached level-0 buffer in the fast mode?  3 processes x (3 hipMalloc, 3 chunk-mapped, 6 uncached)
R=$(pwd); OUT=$R/gpurun_out/r5s13; mkdir -p $OUT
for rep in 1 2 3; do
  timeout 250 $R/build_variants/k1_stream 3 3 0 3 > $OUT/stream$rep.txt 2>&1
  echo "== process $rep"; awk 'NF>20 && ($1 ~ /^[0-9]+$/ || $1=="buf") {print $1, $2, $4, $11, $30, $31}' $OUT/stream$rep.txt | head -13
done
cd $R
for rep in 1 2; do
FVVDP_PLACEMENT_PROBE=1 FVVDP_ALLOC=uncached timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; print('uncached only', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'levels', g['levels_us_per_frame_median'][:3], d['level0_alloc']['in_use'][:12])"
FVVDP_ALLOC=uncached timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; print('uncached x4  ', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'levels', g['levels_us_per_frame_median'][:3], d['level0_alloc']['in_use'][:12], d['level0_alloc']['candidates_us_per_frame'])"
done
