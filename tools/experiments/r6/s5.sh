#!/bin/bash
# round 6, session 5: YUV kernel with 4 waves per workgroup, closed-form sRGB of the 16-bit / float temporal kernels with the wave-uniform
# toe skip: tests, A/B (same box, alternating), then the randomised sweeps against the oracle on this build
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s5
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_state.py tests/test_gpu_sizes.py tests/test_gpu_max_sizes.py tests/test_gpu_cabi_plain.py -x -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
cd /tmp && export TMPDIR=/tmp
SPECS="2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444 2160x3840x60:10:420 1080x1920x60:8:420"
K1S="30:60:u16 60:60:u16 30:60:f32rgb 30:60:f32gray 30:60:u8"
for i in 1 2 3; do
  FVVDP_LIB=$R/build_variants/r6_yuv_step2.so python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn >> $O/yuv_step2.txt
  python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn >> $O/yuv_wpb.txt
  FVVDP_LIB=$R/build_variants/r6_yuv_step2.so python $R/tools/gpu_fps.py $K1S 2>/dev/null | grep -v Warn >> $O/k1_before.txt
  python $R/tools/gpu_fps.py $K1S 2>/dev/null | grep -v Warn >> $O/k1_after.txt
done
python $R/tools/experiments/gpu_stress.py 500 6101 > $O/stress_default.txt 2>&1
FVVDP_BAND_FUSE=1 python $R/tools/experiments/gpu_stress.py 400 6102 > $O/stress_fuse.txt 2>&1
HLO=200 HHI=1100 WLO=300 WHI=2000 NMAX=5 python $R/tools/experiments/gpu_stress.py 200 6103 > $O/stress_mid.txt 2>&1
python $R/tools/experiments/gpu_stress_yuv.py 300 6104 > $O/stress_yuv.txt 2>&1
python $R/tools/experiments/gpu_stress_heat.py 60 6105 > $O/stress_heat.txt 2>&1
python $R/tools/experiments/gpu_stress_shapes.py > $O/stress_shapes.txt 2>&1
tail -n 1 $O/stress_*.txt
