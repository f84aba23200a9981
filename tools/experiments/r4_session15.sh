#!/bin/bash
# round 4, session 15: order in which the physical chunks are mapped into the level-0 range (creation order vs a coprime stride)
R=$(pwd); OUT=$R/gpurun_out/r4s15; mkdir -p $OUT
cd $R
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"])'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
for rep in 1 2 3; do
  for mode in "stride32:FVVDP_VMM_ORDER=stride" "linear32:FVVDP_VMM_ORDER=linear" "stride2:FVVDP_VMM_ORDER=stride FVVDP_VMM_CHUNK_MB=2" "linear2:FVVDP_VMM_ORDER=linear FVVDP_VMM_CHUNK_MB=2" "malloc:FVVDP_ALLOC=malloc"; do
    name=${mode%%:*}; envs=${mode#*:}
    echo -n "$name  " >> $OUT/order.txt
    env $envs python bench.py $B 2>/dev/null | python -c "$P" >> $OUT/order.txt
  done
done
cat $OUT/order.txt
./build_variants/chunks 0 32 2>&1 | tee $OUT/chunks_microbench.txt
