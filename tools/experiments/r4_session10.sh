#!/bin/bash
# round 4, session 10: LDS-staged row prefetch (two steps ahead) in the two-level kernel: parity + same-box A/B
R=$(pwd); OUT=$R/gpurun_out/r4s10; mkdir -p $OUT
cd $R
FVVDP_LIB=$R/build_variants/r4_stage.so timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -m gpu -x -q -k "fused or two_level or full_size or config5 or readme or golden" > $OUT/pytest_stage.txt 2>&1
tail -4 $OUT/pytest_stage.txt
for rep in 1 2 3; do
  echo "== default (registers, one step ahead)" >> $OUT/ab.txt
  timeout 300 python tools/gpu_bandonly_speed.py 12 2>/dev/null | grep -v amdgpu >> $OUT/ab.txt
  echo "== LDS-staged, two steps ahead" >> $OUT/ab.txt
  FVVDP_LIB=$R/build_variants/r4_stage.so timeout 300 python tools/gpu_bandonly_speed.py 12 2>/dev/null | grep -v amdgpu >> $OUT/ab.txt
done
cat $OUT/ab.txt
B="--no-cpu-baseline --no-h2d --no-measure-traffic --steps 20 --warmup 5"
timeout 300 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('default ms', d['ms_per_step'], d['graded_pass']['levels_us_per_frame_median'][:3], d['jod'])"
FVVDP_LIB=$R/build_variants/r4_stage.so timeout 300 python bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('staged ms', d['ms_per_step'], d['graded_pass']['levels_us_per_frame_median'][:3], d['jod'])"
