#!/bin/bash
# round 4, session 33: ticketed temporal kernel on further boxes (short form of session 32)
R=$(pwd); OUT=$R/gpurun_out/r4s33; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"])'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
run() { # label, env, bench args
  echo -n "$1  " >> $OUT/scan.txt
  env $2 python bench.py $B $3 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
}
for rep in 1 2 3; do
  for tk in 1 0; do
    run "4k60 ticket=$tk" "FVVDP_K1_TICKET=$tk" ""
    run "4k60 malloc ticket=$tk" "FVVDP_K1_TICKET=$tk FVVDP_ALLOC=malloc" ""
    run "4k60fps ticket=$tk" "FVVDP_K1_TICKET=$tk" "--fps 60"
  done
done
cat $OUT/scan.txt
