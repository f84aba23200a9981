#!/bin/bash
# round 5, session 14: what the driver runs at round end -- the GPU suite, smoke(), bench.py --gpus 1 --steps 20 --warmup 5
R=$(pwd); OUT=$R/gpurun_out/r5s14; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python -c "
import json; d=json.loads(open('$OUT/bench_driver.json').read().strip().splitlines()[-1]); g=d['graded_pass']
print('driver-style', d['value'], d['ms_per_step'], d['timing'], 'K1', g['temporal_us_per_frame_median'], 'levels', g['levels_us_per_frame_median'], 'graded', g['hbm_frac_all_levels'], 'roof', d['roofline']['frac'], d['roofline_k1']['frac'], d['level0_alloc'], d['cpu_baseline']['value'], d['value_h2d_inclusive'])"
