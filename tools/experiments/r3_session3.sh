#!/bin/bash
# round-3 GPU session 3: tests with the tail launch, A/B of the tail launch in bench.py, K1 level-0 offset sweep, 64-slot ring
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log )
for t in 1 0 1 0; do
  FVVDP_BAND_TAIL=$t timeout 300 python $R/bench.py --no-cpu-baseline --no-h2d --no-measure-traffic --steps 10 > $OUT/bench_tail$t.json 2>/dev/null
  python -c "
import json;d=json.load(open('$OUT/bench_tail$t.json'));g=d['graded_pass'];print('tail=$t ms_per_step',d['ms_per_step'],'levels',g['levels_us_per_frame_median'],'fin',g['finalize_us_per_frame'],'all',g['us_per_frame_all_levels'],'K1',g['temporal_us_per_frame_median'])" >> $OUT/tail_ab.txt
done
timeout 300 python $R/tools/gpu_fps.py 144:120:u8 240:120:u8 120:120:u8 144:60:u16 2>/dev/null | grep -v Warn > $OUT/fps_probe.txt
timeout 200 python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^Q_per|^kernel" | tail -3 > $OUT/fov_probe.txt
timeout 200 python $R/tools/experiments/gpu_image.py > $OUT/image_probe.txt 2>&1
timeout 600 python $R/tools/experiments/gpu_k1_offset_sweep.py > $OUT/k1_offsets.txt 2>&1
ls -la $OUT
