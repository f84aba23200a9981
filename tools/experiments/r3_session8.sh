#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s8
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/tools/experiments/gpu_k1_clocks.py 14 > $OUT/k1_clocks.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ff -o ff -- $R/build_variants/fuse_front 64 14 60 > $OUT/fuse_front.txt 2>/tmp/ff.err
timeout 100 $R/build_variants/fuse_front 32 14 60 >> $OUT/fuse_front.txt 2>&1
timeout 100 $R/build_variants/fuse_front 128 14 60 >> $OUT/fuse_front.txt 2>&1
timeout 100 $R/build_variants/fuse_front 64 0 60 >> $OUT/fuse_front.txt 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/ff -name "*.db" | head -1) > $OUT/fuse_front_trace.md 2>&1
ls -la $OUT
