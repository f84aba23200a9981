#!/bin/bash
# round 4, session 22: waves per workgroup (launch-time FVVDP_BAND2_WPB) over frame sizes, batch lengths and chunk heights
R=$(pwd); OUT=$R/gpurun_out/r4s22; mkdir -p $OUT
cd $R
export FVVDP_LIB=$R/build_variants/wpbrt.so FVVDP_PLACEMENT_PROBE=0
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"], "JOD", d.get("jod"))'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
run() { # label, env, bench args
  echo -n "$1  " >> $OUT/scan.txt
  env $2 python bench.py $B $3 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
}
for rep in 1 2; do
for w in 1 2 4; do
  run "4k60 wpb$w" "FVVDP_BAND2_WPB=$w" ""
done
for kr in 39 54 108; do
  run "4k60 wpb4 kr$kr" "FVVDP_BAND2_WPB=4 FVVDP_BAND2_KR=$kr" ""
  run "4k60 wpb1 kr$kr" "FVVDP_BAND2_WPB=1 FVVDP_BAND2_KR=$kr" ""
done
done
for w in 1 2 3 4; do
  run "fhd60 wpb$w" "FVVDP_BAND2_WPB=$w" "--width 1920 --height 1080 --display standard_fhd"
  run "fhd240 wpb$w" "FVVDP_BAND2_WPB=$w" "--width 1920 --height 1080 --display standard_fhd --frames 240"
done
for w in 1 4; do
  run "4k30 wpb$w" "FVVDP_BAND2_WPB=$w" "--frames 30"
  run "4k120 wpb$w" "FVVDP_BAND2_WPB=$w" "--frames 120"
  run "1440p60 wpb$w" "FVVDP_BAND2_WPB=$w" "--width 2560 --height 1440"
  run "8k16 wpb$w" "FVVDP_BAND2_WPB=$w" "--width 7680 --height 4320 --frames 16"
done
cat $OUT/scan.txt
