#!/bin/bash
# session 20: YUV ingest with the window kept in place (switch over the slot): tests + kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s20
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ky -o yuv -- python $R/tools/gpu_yuv.py 2160x3840x60:8:420 2160x3840x60:10:420:60 2160x3840x60:8:444 2160x3840x60:10:444:60 > $OUT/yuv_probe.txt 2> /tmp/ky.err
python $R/tools/rocpd_summary.py $(find /tmp/ky -name "*.db" | head -1) --only temporal_yuv > $OUT/kernel_trace_yuv.md
cat $OUT/yuv_probe.txt; cat $OUT/kernel_trace_yuv.md | cut -c1-200
