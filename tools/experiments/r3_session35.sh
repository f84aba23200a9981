#!/bin/bash
# session 35: larger randomised sweeps of the final build against the oracle (new seeds)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s35
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( echo "# larger sweeps, new seeds (tools/experiments/r3_session35.sh)"
  timeout 2400 python $R/tools/experiments/gpu_stress.py 1200 9001 2>/dev/null | grep -E "^FAIL|^cases"
  timeout 1200 python $R/tools/experiments/gpu_stress_yuv.py 400 9002 2>/dev/null | grep -E "^FAIL|^cases"
  timeout 1200 python $R/tools/experiments/gpu_stress_heat.py 300 9003 2>/dev/null | grep -E "^FAIL|^heat" ) > $OUT/stress_large.txt
cat $OUT/stress_large.txt
