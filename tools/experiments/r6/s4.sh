#!/bin/bash
# round 6, session 4: the whole GPU suite on the build with the DPP-exchanged chroma columns / fused affine steps of the YUV ingest, the
# choose_level0 changes (original range kept until two halves are in hand, no creation-time timing pass) and the table-range clamp;
# YUV timing against the two earlier builds (same box, alternating); bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s4
mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
SPECS="2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444 2160x3840x60:10:420 1080x1920x60:8:420"
for i in 1 2 3; do
  FVVDP_LIB=$R/build_variants/r6_pre_yuv.so python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn >> $O/yuv_r5.txt
  FVVDP_LIB=$R/build_variants/r6_yuv_step1.so python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn >> $O/yuv_step1.txt
  python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn >> $O/yuv_step2.txt
done
YUV_DARK=1 python $R/tools/gpu_yuv.py 2160x3840x60:8:420 2>/dev/null | grep -v Warn > $O/yuv_dark_step2.txt
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
rocprofv3 --pmc $SQ1 --kernel-trace -d /tmp/sq_a -o a -- python $R/tools/gpu_yuv.py 2160x3840x60:8:420 2160x3840x60:10:420:60 > /tmp/sq_a.log 2>&1
python $R/tools/pmc_sq_summary.py temporal_yuv $(find /tmp/sq_a -name "*.db") > $O/pmc_sq_yuv_step2.md 2>/dev/null
rocprofv3 --kernel-trace --stats -d /tmp/ky -o yuv -- python $R/tools/gpu_yuv.py $SPECS > /dev/null 2> /tmp/ky.err
python $R/tools/rocpd_summary.py $(find /tmp/ky -name "*.db" | head -1) --only temporal_yuv > $O/kernel_trace_yuv.md
python $R/bench.py --no-cpu-baseline --no-h2d > $O/bench.json 2> $O/bench.err
ls $O
