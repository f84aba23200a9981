#!/bin/bash
# round 6, session 8: the YUV kernel's floors and knobs on the current build -- arithmetic alone (YUV_ABLATE_MEM), one frame of raw samples
# in flight instead of two (YUV_TD8=1), 4 waves per SIMD asked for (YUV_WAVES8=4: spills); same box, alternating processes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s8
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SPECS="2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444"
for i in 1 2 3; do
  for v in base ${VARIANTS:-yuv_ablmem yuv_td1 yuv_w4}; do
    if [ $v = base ]; then L=""; else L=$R/build_variants/$v.so; fi
    FVVDP_LIB=$L python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn | sed "s/^/$v #$i /" >> $O/yuv.txt
  done
done
cat $O/yuv.txt
