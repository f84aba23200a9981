#!/bin/bash
# round 4, session 26: short chunks dispatched last (two-phase band2 launch), FVVDP_BAND2_KR2 = rows of the short chunks (0 = uniform)
R=$(pwd); OUT=$R/gpurun_out/r4s26; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0 FVVDP_LIB=$R/build_variants/tail2.so
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "not foveated" > $OUT/tests.txt 2>&1
tail -n 3 $OUT/tests.txt
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"], "JOD", d.get("jod"))'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
run() { # label, env, bench args
  echo -n "$1  " >> $OUT/scan.txt
  env $2 python bench.py $B $3 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
}
for rep in 1 2; do
  for k2 in 0 20 10 13 26 39; do
    run "4k60 kr2=$k2" "FVVDP_BAND2_KR2=$k2" ""
  done
  run "4k60 kr2=20 wpb1" "FVVDP_BAND2_KR2=20 FVVDP_BAND2_WPB=1" ""
  run "4k60 kr2=0 wpb1" "FVVDP_BAND2_KR2=0 FVVDP_BAND2_WPB=1" ""
done
for k2 in 0 default; do
  E="FVVDP_BAND2_KR2=$k2"; [ $k2 = default ] && E="X=1"
  run "fhd60 kr2=$k2" "$E" "--width 1920 --height 1080 --display standard_fhd"
  run "4k120 kr2=$k2" "$E" "--frames 120"
  run "4k30 kr2=$k2" "$E" "--frames 30"
  run "1440p kr2=$k2" "$E" "--width 2560 --height 1440"
  run "8k16 kr2=$k2" "$E" "--width 7680 --height 4320 --frames 16"
done
cat $OUT/scan.txt
FVVDP_LIB=$R/build_variants/timeline2.so python tools/gpu_timeline.py > $OUT/timeline_4k.txt 2>>$OUT/err.txt
cat $OUT/timeline_4k.txt
