#!/bin/bash
# round 6, session 3: YUV ingest after the matrix / sRGB-branch change: parity tests, A/B against the build before it (same box, alternating),
# dark content, dynamic instruction counts (SQ counters) of the 8-bit 4:2:0 kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s3
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "yuv" > $O/pytest_yuv.log 2>&1
echo "pytest rc $?" >> $O/pytest_yuv.log
cd /tmp && export TMPDIR=/tmp
SPECS="2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444 2160x3840x60:10:420"
for i in 1 2 3; do
  FVVDP_LIB=$R/build_variants/r6_pre_yuv.so python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn >> $O/yuv_before.txt
  python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn >> $O/yuv_after.txt
done
YUV_DARK=1 FVVDP_LIB=$R/build_variants/r6_pre_yuv.so python $R/tools/gpu_yuv.py 2160x3840x60:8:420 2>/dev/null | grep -v Warn > $O/yuv_dark_before.txt
YUV_DARK=1 python $R/tools/gpu_yuv.py 2160x3840x60:8:420 2>/dev/null | grep -v Warn > $O/yuv_dark_after.txt
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
for T in before after; do
  L=$R/fovvideovdp_amd/libfvvdp_hip.so; [ $T = before ] && L=$R/build_variants/r6_pre_yuv.so
  FVVDP_LIB=$L rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/sq_$T -o a -- python $R/tools/gpu_yuv.py 2160x3840x60:8:420 > /tmp/sq_$T.log 2>&1
  python $R/tools/pmc_sq_summary.py temporal_yuv $(find /tmp/sq_$T -name "*.db") > $O/pmc_sq_yuv_$T.md 2>/dev/null
done
rocprofv3 --kernel-trace --stats -d /tmp/ky -o yuv -- python $R/tools/gpu_yuv.py $SPECS > /dev/null 2> /tmp/ky.err
python $R/tools/rocpd_summary.py $(find /tmp/ky -name "*.db" | head -1) --only temporal_yuv > $O/kernel_trace_yuv.md
ls $O
