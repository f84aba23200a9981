#!/bin/bash
# round 6, session 16: the YUV ingest behind the PQ display model (HDR10-like: 10-bit 4:2:0) with the packed PQ of the 16-bit / float kernels instead
# of the per-component form: YUV tests, then A/B on one box, alternating processes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s16
mkdir -p $O
cd $R
for v in r6_pre_yuvpq yuv_pq; do
  FVVDP_LIB=$R/build_variants/$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_sizes.py -q -m gpu -k "yuv" > $O/pytest_$v.log 2>&1
  echo "$v pytest rc $?"; tail -1 $O/pytest_$v.log
done
cd /tmp && export TMPDIR=/tmp
SPECS="2160x3840x60:10:420 2160x3840x60:10:420:60 2160x3840x60:8:420 2160x3840x60:10:444 1080x1920x60:10:420"
for i in 1 2 3; do
  for v in r6_pre_yuvpq yuv_pq; do
    FVVDP_LIB=$R/build_variants/$v.so PROBE_DISPLAY=standard_hdr_pq python $R/tools/gpu_yuv.py $SPECS 2>/dev/null | grep -v Warn | sed "s/^/$v #$i /" >> $O/yuv.txt
  done
done
python - $O <<'PY'
import sys,re,collections,statistics as st
d=collections.defaultdict(list)
for l in open(sys.argv[1]+"/yuv.txt"):
    m=re.match(r"^(\S+) #\d \S+ (\S+) .*JOD ([\d.]+) .*temporal ([\d.]+)",l)
    if m: d[(m.group(2),m.group(1))].append((float(m.group(4)),m.group(3)))
for k in sorted(d): print(k[0],k[1],[x[0] for x in d[k]],"median",st.median(x[0] for x in d[k]),"JOD",d[k][0][1])
PY
