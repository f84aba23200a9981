#!/bin/bash
# session 30: automatic tail launch (levels holding <= FVVDP_TAIL_PX pixels per frame + finalize + pooling in one launch)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s30
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PX in 0 2048 8192 40000; do
  export FVVDP_TAIL_PX=$PX
  echo "== FVVDP_TAIL_PX=$PX" | tee -a $OUT/tail_auto.txt
  python $R/tools/gpu_small_latency.py 512x512 1080x1920 2160x3840 512x512x12 1080x1920x12 2>&1 | grep -v Warn | grep -v amdgpu | sed 's/sync=False: host [0-9.]* us.call, //' | tee -a $OUT/tail_auto.txt
  for rep in 1 2; do python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench ms_per_step', d['ms_per_step'], 'levels', d['graded_pass']['levels_us_per_frame_median'], 'fin', d['graded_pass']['finalize_us_per_frame'])" | tee -a $OUT/tail_auto.txt; done
  python $R/tools/experiments/gpu_speed.py 1080x1920x60 2>/dev/null | grep -v Warn | tee -a $OUT/tail_auto.txt
done
