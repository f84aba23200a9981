#!/bin/bash
# round 5, session 21: which of the forced-split failures depend on the kind of the second range / on what the ranges hold before the first write
R=$(pwd); OUT=$R/gpurun_out/r5s21; mkdir -p $OUT
cd $R
K='waves_per_workgroup_change_no_bits or band2_tickets_change_no_bits or pipelined_source_feeder'
for rep in 1 2; do
for v in "FVVDP_LEVEL0_SPLIT=1" "FVVDP_LEVEL0_SPLIT=1 FVVDP_ALLOC=malloc" "FVVDP_LEVEL0_SPLIT=2" "FVVDP_LEVEL0_SPLIT=0"; do
  echo "== $v"; env $v timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_state.py -m gpu -q -k "$K" 2>&1 | grep -E "^FAILED|passed|failed" 
done; done
