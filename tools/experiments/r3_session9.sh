#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s10
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FVVDP_PLACEMENT_PROBE=1 FVVDP_DEBUG_ALLOC=1 timeout 600 python $R/tools/experiments/gpu_k1_probe_check.py 8 > $OUT/probe_on.txt 2>&1
FVVDP_PLACEMENT_PROBE=0 timeout 600 python $R/tools/experiments/gpu_k1_probe_check.py 8 > $OUT/probe_off.txt 2>&1
for rep in 1 2 3; do
  FVVDP_PLACEMENT_PROBE=1 timeout 300 python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic --warmup 2 --steps 10 > $OUT/b_on$rep.json 2>/dev/null
  FVVDP_PLACEMENT_PROBE=0 timeout 300 python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic --warmup 2 --steps 10 > $OUT/b_off$rep.json 2>/dev/null
  for t in on off; do python -c "
import json;d=json.load(open('$OUT/b_$t$rep.json'));g=d['graded_pass'];print('probe=$t ms_per_step',d['ms_per_step'],'K1',g['temporal_us_per_frame_median'],'K2b',g['levels_us_per_frame_median'][0],'all',g['us_per_frame_all_levels'])" >> $OUT/bench_ab.txt; done
done
( cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log )
ls -la $OUT
