#!/bin/bash
# round-3 GPU session 4: tests, tail-launch A/B (tiny levels only), K1 address sweep, YUV probe
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log )
for t in d 0 d 0; do
  if [ $t = d ]; then unset FVVDP_BAND_TAIL; else export FVVDP_BAND_TAIL=$t; fi
  timeout 300 python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic --steps 10 > $OUT/bench_tail$t.json 2>/dev/null
  python -c "
import json;d=json.load(open('$OUT/bench_tail$t.json'));g=d['graded_pass'];print('tail=$t ms_per_step',d['ms_per_step'],'levels',g['levels_us_per_frame_median'],'fin',g['finalize_us_per_frame'],'all',g['us_per_frame_all_levels'],'K1',g['temporal_us_per_frame_median'])" >> $OUT/tail_ab.txt
done
unset FVVDP_BAND_TAIL
timeout 200 python $R/tools/experiments/gpu_image.py 2>&1 | grep image > $OUT/image_probe.txt
FVVDP_BAND_TAIL=0 timeout 200 python $R/tools/experiments/gpu_image.py 2>&1 | grep image | sed 's/^/tail=0 /' >> $OUT/image_probe.txt
timeout 300 python $R/tools/gpu_yuv.py 2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444 > $OUT/yuv_probe.txt 2>/dev/null
timeout 900 python $R/tools/experiments/gpu_k1_offset_sweep.py > $OUT/k1_offsets.txt 2>&1
ls -la $OUT
