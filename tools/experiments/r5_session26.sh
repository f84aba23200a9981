#!/bin/bash
# round 5, session 26: does K1 drift with the temperature of the memory?  twenty 4K bench processes back to back, the card's sensors read before each
R=$(pwd); OUT=$R/gpurun_out/r5s26; mkdir -p $OUT
cd $R
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['graded_pass']; a=d['level0_alloc']; k=d['roofline_k1']; print('$1', d['ms_per_step'], 'K1', g['temporal_us_per_frame_median'], 'min/max', round(k['min_launch_ms']*1000/60,2), round(k['max_launch_ms']*1000/60,2), 'lv01', g['levels_us_per_frame_median'][0], '|', a['kept_indices'], a['pair_write_rate_tbs'], a['temporal_plus_pyramid_us_per_frame_at_creation'])"; }
sens() { /opt/rocm/bin/rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -E "Temperature|Power|sclk|mclk|fclk" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';' | cut -c1-400; echo; }
/opt/rocm/bin/rocm-smi --showtemp --showpower --showclocks 2>&1 | head -40 > $OUT/smi_full.txt
for rep in $(seq 1 20); do
  sens
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | line "run $rep"
done > $OUT/drift.txt 2>&1
sens >> $OUT/drift.txt
cat $OUT/drift.txt
