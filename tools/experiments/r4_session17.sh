#!/bin/bash
# round 4, session 17: online choice between the two level-0 allocation kinds -- does it pick the faster one on this box?
R=$(pwd); OUT=$R/gpurun_out/r4s17; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_state.py -x -q -m gpu -k "allocation_choice or chunk_mapped or stage_overlap" > $OUT/tests.txt 2>&1
tail -n 5 $OUT/tests.txt
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; a=d["level0_alloc"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "|", a["start"], "->", a["in_use"], a["compared_us_per_frame"], a["kept"])'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
for rep in 1 2 3; do
  for mode in "choice:X=1" "vmm_fixed:FVVDP_PLACEMENT_PROBE=0" "malloc_fixed:FVVDP_ALLOC=malloc FVVDP_PLACEMENT_PROBE=0" "choice_from_malloc:FVVDP_ALLOC=malloc"; do
    name=${mode%%:*}; envs=${mode#*:}
    echo -n "$name  " >> $OUT/choice.txt
    env $envs python bench.py $B 2>$OUT/err_$name.txt | python -c "$P" >> $OUT/choice.txt 2>&1 || echo failed >> $OUT/choice.txt
  done
done
cat $OUT/choice.txt
