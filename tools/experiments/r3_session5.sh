#!/bin/bash
# round-3 GPU session 5: tail launch with 4 waves vs 16 vs none; K1 in-allocation offset sweep
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for t in wpt4 wpt16 none; do
  unset FVVDP_BAND_TAIL FVVDP_LIB
  [ $t = none ] && export FVVDP_BAND_TAIL=0
  [ $t = wpt16 ] && export FVVDP_LIB=$R/build_variants/tail16.so
  timeout 300 python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic --steps 10 > $OUT/bench_$t.json 2>/dev/null
  python -c "
import json;d=json.load(open('$OUT/bench_$t.json'));g=d['graded_pass'];print('tail=$t ms_per_step',d['ms_per_step'],'levels',g['levels_us_per_frame_median'],'fin',g['finalize_us_per_frame'],'all',g['us_per_frame_all_levels'],'K1',g['temporal_us_per_frame_median'])" >> $OUT/tail_ab.txt
done
done
unset FVVDP_BAND_TAIL FVVDP_LIB
timeout 200 python $R/tools/experiments/gpu_image.py 2>&1 | grep image > $OUT/image_probe.txt
FVVDP_BAND_TAIL=0 timeout 200 python $R/tools/experiments/gpu_image.py 2>&1 | grep image | sed 's/^/tail=0 /' >> $OUT/image_probe.txt
timeout 900 python $R/tools/experiments/gpu_k1_offset_sweep.py 5 > $OUT/k1_offsets.txt 2>&1
ls -la $OUT
