#!/bin/bash
# round 6, session 13: the 16-bit / float RGB temporal kernels with the next frame requested BEFORE the current one is converted
# (-DK1_EARLY_WORDS=4; 125 -> 135 / 130 -> 146 registers, still 3 waves per SIMD): A/B, same box, alternating processes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6s13
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  python $R/tools/gpu_fps.py 30:60:u16 60:60:u16 30:60:f32rgb 30:60:f32gray 2>/dev/null | grep -v Warn | sed "s/^/base #$i /" >> $O/k1.txt
  FVVDP_LIB=$R/build_variants/k1_early_u16.so python $R/tools/gpu_fps.py 30:60:u16 60:60:u16 2>/dev/null | grep -v Warn | sed "s/^/early #$i /" >> $O/k1.txt
  FVVDP_LIB=$R/build_variants/k1_early_f32.so python $R/tools/gpu_fps.py 30:60:f32rgb 30:60:f32gray 2>/dev/null | grep -v Warn | sed "s/^/early #$i /" >> $O/k1.txt
done
sort -k4,8 -k1,1 $O/k1.txt
