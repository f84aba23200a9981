#!/bin/bash
# round 5, session 30: the card's power and clocks WHILE the step runs back to back (is the steady state of the long launches a power cap?)
R=$(pwd); OUT=$R/gpurun_out/r5s30; mkdir -p $OUT
cd $R
/opt/rocm/bin/rocm-smi --showmaxpower --showpower --showclocks --showperflevel 2>&1 | grep -v "^=\|^$" > $OUT/idle.txt
timeout 600 python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic --steps 4000 --warmup 3 > $OUT/bench_long.json 2>/dev/null &
BP=$!
sleep 3
for k in $(seq 1 30); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|junction|memory" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'; echo
  sleep 0.5
done > $OUT/load.txt
wait $BP
python -c "
import json; d=json.loads(open('$OUT/bench_long.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'mean', d['timing']['ms_per_step_mean'], 'min', d['timing']['ms_per_step_min'], 'max', d['timing']['ms_per_step_max'])"
cat $OUT/idle.txt; cat $OUT/load.txt
