#!/bin/bash
# chunk height of the two-level kernel (level-C rows per chunk) at 4K x60: the cost model's choice against fixed values
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/s32
cd /tmp
for KR in auto 20 27 34 45 54 68 90 135 270 540; do
  if [ $KR = auto ]; then unset FVVDP_BAND2_KR; else export FVVDP_BAND2_KR=$KR; fi
  python $R/tools/gpu_bandonly_speed.py 12 2>/dev/null | grep -v Warn | tail -1 | sed "s/^.*bands us/kr=$KR bands us/" | cut -c1-110 | tee -a $R/gpurun_out/s32/kr_sweep.txt
done
