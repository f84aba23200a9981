#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s13
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log )
timeout 300 python $R/tools/gpu_fps.py 144:60:f32gray 240:60:f32gray 144:60:u16 144:60:f32rgb 2>/dev/null | grep -v Warn > $OUT/fps_probe.txt
ls -la $OUT
