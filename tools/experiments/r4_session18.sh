#!/bin/bash
# round 4, session 18: the allocation choice on further boxes (shorter form of session 17)
R=$(pwd); OUT=$R/gpurun_out/r4s18; mkdir -p $OUT
cd $R
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; a=d["level0_alloc"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "|", a["start"], "->", a["in_use"], a["compared_us_per_frame"], a["kept"])'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/choice.txt
for rep in 1 2; do
  for mode in "choice:X=1" "vmm_fixed:FVVDP_PLACEMENT_PROBE=0" "malloc_fixed:FVVDP_ALLOC=malloc FVVDP_PLACEMENT_PROBE=0" "choice_from_malloc:FVVDP_ALLOC=malloc"; do
    name=${mode%%:*}; envs=${mode#*:}
    echo -n "$name  " >> $OUT/choice.txt
    env $envs python bench.py $B 2>$OUT/err_$name.txt | python -c "$P" >> $OUT/choice.txt 2>&1 || echo failed >> $OUT/choice.txt
  done
done
cat $OUT/choice.txt
