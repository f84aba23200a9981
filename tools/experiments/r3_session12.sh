#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s12
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( echo "# Round 03 -- randomised parity sweeps of the final build against the oracle on MI355X (tools/experiments/gpu_stress*.py; one gpurun call)"
  echo "# gpu_stress.py 300 303: sizes 17..150 x 17..260, 1..13 frames, 0/24/25/30/50/60/120/144/240 fps, three paddings, uint8/uint16/float,"
  echo "#   RGB/gray, standard_4k/fhd/hdr_pq/hmd, foveated 1 in 4 (fail = |dJOD| > 5e-4, Q_per_ch > 2e-2 relative, or an exception on one side only)"
  timeout 1500 python $R/tools/experiments/gpu_stress.py 300 303 2>/dev/null | tail -1
  echo "# gpu_stress_yuv.py 120 78: raw planar YUV ingest, 8..16 bit, 4:2:0 / 4:4:4, bt709 / bt2020, several frame rates and displays"
  timeout 900 python $R/tools/experiments/gpu_stress_yuv.py 120 78 2>/dev/null | tail -1
  echo "# gpu_stress_shapes.py: very wide / tall / tiny frames and sizes around the strip and tile boundaries of the kernels"
  timeout 900 python $R/tools/experiments/gpu_stress_shapes.py 2>/dev/null | tail -1 ) > $OUT/stress.txt
cat $OUT/stress.txt
