#!/bin/bash
# round 4, session 28: resident workgroups taking items from per-XCD queues (FVVDP_BAND2_QUEUE=0: one workgroup per item)
R=$(pwd); OUT=$R/gpurun_out/r4s28; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "not foveated" > $OUT/tests.txt 2>&1
tail -n 3 $OUT/tests.txt
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "levels", g["levels_us_per_frame_median"][:4], "all", g["us_per_frame_all_levels"], "JOD", d.get("jod"))'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
run() { # label, env, bench args
  echo -n "$1  " >> $OUT/scan.txt
  env $2 python bench.py $B $3 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
}
for rep in 1 2 3; do
  run "4k60 queues" "X=1" ""
  run "4k60 hw-dispatch" "FVVDP_BAND2_QUEUE=0" ""
  run "4k60 hw-dispatch (previous build)" "FVVDP_LIB=$R/build_variants/tail2.so" ""
done
for q in 1 0; do
  run "fhd60 queue=$q" "FVVDP_BAND2_QUEUE=$q" "--width 1920 --height 1080 --display standard_fhd"
  run "4k120 queue=$q" "FVVDP_BAND2_QUEUE=$q" "--frames 120"
  run "1440p queue=$q" "FVVDP_BAND2_QUEUE=$q" "--width 2560 --height 1440"
  run "8k16 queue=$q" "FVVDP_BAND2_QUEUE=$q" "--width 7680 --height 4320 --frames 16"
done
cat $OUT/scan.txt
FVVDP_LIB=$R/build_variants/timeline3.so python tools/gpu_timeline.py > $OUT/timeline_4k.txt 2>>$OUT/err.txt
cat $OUT/timeline_4k.txt
