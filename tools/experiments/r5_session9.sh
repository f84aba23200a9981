#!/bin/bash
# round 5, session 9: GPU test suite (all of it), then same-box A/B of the round-5 kernel changes
R=$(pwd); OUT=$R/gpurun_out/r5s9; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log
for rep in 1 2; do
  FVVDP_PLACEMENT_PROBE=0 python tools/gpu_config4.py 2>/dev/null | grep "kernel us/frame:" | sed "s/^/fin=on  /"
  FVVDP_PLACEMENT_PROBE=0 FVVDP_BAND_INRANGE=0 python tools/gpu_config4.py 2>/dev/null | grep "kernel us/frame:" | sed "s/^/fin=off /"
done > $OUT/fov_ab.txt
cat $OUT/fov_ab.txt
