#!/bin/bash
# round 4, session 42: a chip-wide metronome for the stages of band2_kernel (every wave starts a stage on a multiple of T ticks of the 100 MHz clock)
R=$(pwd); OUT=$R/gpurun_out/r4s42; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0 FVVDP_LIB=$R/build_variants/tslot.so
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"], "JOD", d.get("jod"))'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
run() { # label, env, bench args
  echo -n "$1  " >> $OUT/scan.txt
  env $2 python bench.py $B $3 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
}
for T in 0 360 380 400 420 440 460 500; do
  run "4k60 T=$T" "FVVDP_BAND2_TSLOT=$T" ""
done
for T in 400 440; do
  run "4k60 T=$T tol=100" "FVVDP_BAND2_TSLOT=$T FVVDP_BAND2_TSLOT_TOL=100" ""
done
cat $OUT/scan.txt
FVVDP_BAND2_TSLOT=420 FVVDP_LIB=$R/build_variants/tslot_tl.so python tools/gpu_timeline.py > $OUT/timeline_t420.txt 2>>$OUT/err.txt
head -8 $OUT/timeline_t420.txt
