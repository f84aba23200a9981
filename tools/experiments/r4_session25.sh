#!/bin/bash
# round 4, session 25: strip <-> wave rotation inside a group (partly filled last groups)
R=$(pwd); OUT=$R/gpurun_out/r4s25; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"], "JOD", d.get("jod"))'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
run() { # label, env, bench args
  echo -n "$1  " >> $OUT/scan.txt
  env $2 python bench.py $B $3 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
}
for rep in 1 2; do
for wh in "3840 2160 standard_4k" "1920 1080 standard_fhd" "3200 1800 standard_4k" "2048 1080 standard_4k" "2560 1440 standard_4k"; do
  set -- $wh
  run "$1x$2 wpb1" "FVVDP_LIB=$R/build_variants/wpbrot.so FVVDP_BAND2_WPB=1" "--width $1 --height $2 --display $3"
  run "$1x$2 wpb4 fixed" "FVVDP_LIB=$R/build_variants/wpbrt.so FVVDP_BAND2_WPB=4" "--width $1 --height $2 --display $3"
  run "$1x$2 wpb4 rotated" "FVVDP_LIB=$R/build_variants/wpbrot.so FVVDP_BAND2_WPB=4" "--width $1 --height $2 --display $3"
  run "$1x$2 wpb2 rotated" "FVVDP_LIB=$R/build_variants/wpbrot.so FVVDP_BAND2_WPB=2" "--width $1 --height $2 --display $3"
done
done
cat $OUT/scan.txt
