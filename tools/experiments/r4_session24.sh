#!/bin/bash
# round 4, session 24: 8 waves per workgroup; two barriers per stage
R=$(pwd); OUT=$R/gpurun_out/r4s24; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"], "JOD", d.get("jod"))'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
run() { # label, env, bench args
  echo -n "$1  " >> $OUT/scan.txt
  env $2 python bench.py $B $3 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
}
for rep in 1 2 3; do
  run "4k60 wpb1" "FVVDP_LIB=$R/build_variants/wpbrt.so FVVDP_BAND2_WPB=1" ""
  run "4k60 wpb4" "FVVDP_LIB=$R/build_variants/wpbrt.so FVVDP_BAND2_WPB=4" ""
  run "4k60 wpb4 sync2" "FVVDP_LIB=$R/build_variants/sync2.so FVVDP_BAND2_WPB=4" ""
  run "4k60 wpb8" "FVVDP_LIB=$R/build_variants/wpbmax8.so FVVDP_BAND2_WPB=8" ""
  run "4k60 wpb4(max8 build)" "FVVDP_LIB=$R/build_variants/wpbmax8.so FVVDP_BAND2_WPB=4" ""
done
run "4k120 wpb1" "FVVDP_LIB=$R/build_variants/wpbrt.so FVVDP_BAND2_WPB=1" "--frames 120"
run "4k120 wpb4" "FVVDP_LIB=$R/build_variants/wpbrt.so FVVDP_BAND2_WPB=4" "--frames 120"
run "4k120 wpb8" "FVVDP_LIB=$R/build_variants/wpbmax8.so FVVDP_BAND2_WPB=8" "--frames 120"
for wh in "2048 1080" "2560 1440" "3200 1800" "2880 1620" "2400 1350"; do
  set -- $wh
  for w in 1 4; do
    run "$1x$2 wpb$w" "FVVDP_LIB=$R/build_variants/wpbrt.so FVVDP_BAND2_WPB=$w" "--width $1 --height $2"
  done
done
cat $OUT/scan.txt
