#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s15
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log )
for rep in 1 2 3; do timeout 200 python $R/tools/gpu_config4.py 2>/dev/null | grep -E "^config4|^Q_per|^kernel" | tail -3 >> $OUT/fov_probe.txt; done
tail -3 $OUT/pytest.log; cat $OUT/fov_probe.txt
