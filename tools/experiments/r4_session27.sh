#!/bin/bash
# round 4, session 27: the mid-size levels (2, 3) through the two-level kernel (FVVDP_BAND_FUSE=1 forces it wherever valid)
R=$(pwd); OUT=$R/gpurun_out/r4s27; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0 FVVDP_LIB=$R/build_variants/tail2.so
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "levels", g["levels_us_per_frame_median"], "all", g["us_per_frame_all_levels"])'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
run() { # label, env, bench args
  echo -n "$1  " >> $OUT/scan.txt
  env $2 python bench.py $B $3 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
}
for rep in 1 2; do
  run "4k60 default" "X=1" ""
  run "4k60 fuse-all" "FVVDP_BAND_FUSE=1" ""
  run "fhd60 default" "X=1" "--width 1920 --height 1080 --display standard_fhd"
  run "fhd60 fuse-all" "FVVDP_BAND_FUSE=1" "--width 1920 --height 1080 --display standard_fhd"
done
cat $OUT/scan.txt
