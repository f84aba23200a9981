#!/bin/bash
# round 6, session 11: the YUV vector kernel with 2 pixels per lane (half the window registers: 4-5 waves per SIMD on the 8-slot window, 3-4 on
# the 16-slot one) against 4 pixels per lane: YUV tests on each build, then A/B on one box, alternating processes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s11
mkdir -p $O
cd $R
for v in ${LIBS:-yuv_px4 yuv_px2 yuv_px2w}; do
  FVVDP_LIB=$R/build_variants/$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_sizes.py -q -m gpu -k "yuv" > $O/pytest_$v.log 2>&1
  echo "$v pytest rc $?" | tee -a $O/pytest_$v.log; tail -2 $O/pytest_$v.log
done
cd /tmp && export TMPDIR=/tmp
SPECS="2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444 2160x3840x60:10:420 2160x3840x60:8:420:60 1080x1920x60:8:420"
for i in 1 2 3; do
  for v in ${LIBS:-yuv_px4 yuv_px2 yuv_px2w}; do
    FVVDP_LIB=$R/build_variants/$v.so python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn | sed "s/^/$v #$i /" >> $O/yuv.txt
  done
done
python - $O <<'PY'
import sys,re,collections,statistics as st
O=sys.argv[1]
d=collections.defaultdict(list)
for l in open(O+"/yuv.txt"):
    m=re.match(r"^(\S+) #\d \S+ (\S+) .*JOD ([\d.]+) .*temporal ([\d.]+)",l)
    if m: d[(m.group(2),m.group(1))].append((float(m.group(4)),m.group(3)))
for k in sorted(d): print(k[0],k[1],[x[0] for x in d[k]],"median",st.median(x[0] for x in d[k]),"JOD",d[k][0][1])
PY
