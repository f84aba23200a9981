#!/bin/bash
# round 4, session 16: does the placement of the SOURCE arrays (torch allocations) decide the temporal kernel's mode?
R=$(pwd); OUT=$R/gpurun_out/r4s16; mkdir -p $OUT
cd $R
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"])'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
for rep in 1 2; do
  for mode in "default:X=1" "src_expandable:PYTORCH_HIP_ALLOC_CONF=expandable_segments:True" "src_expandable_cuda:PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True" "l0_malloc:FVVDP_ALLOC=malloc" "both:FVVDP_ALLOC=malloc PYTORCH_HIP_ALLOC_CONF=expandable_segments:True"; do
    name=${mode%%:*}; envs=${mode#*:}
    echo -n "$name  " >> $OUT/src.txt
    env $envs python bench.py $B 2>$OUT/err_$name.txt | python -c "$P" >> $OUT/src.txt 2>&1 || echo failed >> $OUT/src.txt
  done
done
cat $OUT/src.txt
./build_variants/chunks 0 32 2>&1 | head -2
