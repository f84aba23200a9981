#!/bin/bash
# round 4, session 37: static weighting of the XCDs for the temporal kernel (every k-th workgroup on XCD 1 / 5 idle) vs tickets vs plain
R=$(pwd); OUT=$R/gpurun_out/r4s37; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0 FVVDP_LIB=$R/build_variants/xcdskip.so
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"])'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
run() { # label, env, bench args
  echo -n "$1  " >> $OUT/scan.txt
  env $2 python bench.py $B $3 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
}
for rep in 1 2 3; do
  run "4k60 tickets" "X=1" ""
  run "4k60 plain" "FVVDP_K1_TICKET=0" ""
  for k in 24 16 12 8; do
    run "4k60 skip=$k" "FVVDP_K1_TICKET=0 FVVDP_K1_XCD_SKIP=$k" ""
  done
done
run "4k60fps tickets" "X=1" "--fps 60"
run "4k60fps plain" "FVVDP_K1_TICKET=0" "--fps 60"
run "4k60fps skip=16" "FVVDP_K1_TICKET=0 FVVDP_K1_XCD_SKIP=16" "--fps 60"
run "4k60fps skip=12" "FVVDP_K1_TICKET=0 FVVDP_K1_XCD_SKIP=12" "--fps 60"
cat $OUT/scan.txt
