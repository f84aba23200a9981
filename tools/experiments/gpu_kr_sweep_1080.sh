#!/bin/bash
# chunk height of the two-level kernel at 1080p x60 (135 level-C rows): the cost model's choice against fixed values
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/s34
cd /tmp
export HH=1080 WW=1920 NN=60
for KR in auto 9 12 14 15 17 20 23 27 34 45 68 135; do
  if [ $KR = auto ]; then unset FVVDP_BAND2_KR; else export FVVDP_BAND2_KR=$KR; fi
  python $R/tools/gpu_bandonly_speed.py 12 2>/dev/null | grep -v Warn | tail -1 | sed "s/^.*bands us/kr=$KR bands us/" | cut -c1-110 | tee -a $R/gpurun_out/s34/kr_sweep_1080.txt
done
