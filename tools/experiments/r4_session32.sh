#!/bin/bash
# round 4, session 32: the uint8 temporal kernel as resident workgroups taking pixel blocks from a ticket counter (FVVDP_K1_TICKET=0: off)
R=$(pwd); OUT=$R/gpurun_out/r4s32; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_state.py -x -q -m gpu > $OUT/tests.txt 2>&1
tail -n 3 $OUT/tests.txt
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"], "JOD", d.get("jod"))'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
run() { # label, env, bench args
  echo -n "$1  " >> $OUT/scan.txt
  env $2 python bench.py $B $3 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
}
for rep in 1 2 3; do
  for al in "X=1" "FVVDP_ALLOC=malloc"; do
    run "4k60 tickets $al" "$al" ""
    run "4k60 hw-dispatch $al" "$al FVVDP_K1_TICKET=0" ""
  done
done
for tk in 1 0; do
  run "fhd60 ticket=$tk" "FVVDP_K1_TICKET=$tk" "--width 1920 --height 1080 --display standard_fhd"
  run "4k60fps ticket=$tk" "FVVDP_K1_TICKET=$tk" "--fps 60"
  run "4k120 ticket=$tk" "FVVDP_K1_TICKET=$tk" "--frames 120"
done
cat $OUT/scan.txt
