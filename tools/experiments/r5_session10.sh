#!/bin/bash
# round 5, session 10: per-XCD item counters of the two-level pyramid kernel (FVVDP_BAND2_TICKET=1) against the static split (=0), same
# box, alternating processes; parity of the fused kernels first
R=$(pwd); OUT=$R/gpurun_out/r5s10; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for rep in 1 2 3; do
  for T in 0 1; do
    FVVDP_BAND2_TICKET=$T FVVDP_PLACEMENT_PROBE=0 python tools/gpu_bandonly_speed.py 12 2>/dev/null | tail -2 | sed "s/^/ticket=$T 4K /"
  done
done > $OUT/ab.txt
for T in 0 1; do
  FVVDP_BAND2_TICKET=$T python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ticket=$T bench', d['ms_per_step'], d['graded_pass']['levels_us_per_frame_median'], d['roofline_k1']['median_launch_ms'], d['jod'])"
  FVVDP_BAND2_TICKET=$T python bench.py --width 1920 --height 1080 --display standard_fhd --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ticket=$T fhd  ', d['ms_per_step'], d['graded_pass']['levels_us_per_frame_median'], d['roofline_k1']['median_launch_ms'], d['jod'])"
done >> $OUT/ab.txt
cat $OUT/ab.txt
