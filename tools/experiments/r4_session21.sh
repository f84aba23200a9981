#!/bin/bash
# round 4, session 21: more waves per workgroup in step (6, 9, 12 of the 36 strips of a 4K row)
R=$(pwd); OUT=$R/gpurun_out/r4s21; mkdir -p $OUT
cd $R
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"], "JOD", d.get("jod"))'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/ab.txt
for rep in 1 2 3; do
  for v in wpb1 wpb4 wpb6 wpb9 wpb12; do
    echo -n "$v  " >> $OUT/ab.txt
    FVVDP_PLACEMENT_PROBE=0 FVVDP_LIB=$R/build_variants/$v.so python bench.py $B 2>$OUT/err_$v.txt | python -c "$P" >> $OUT/ab.txt 2>&1 || echo failed >> $OUT/ab.txt
  done
done
for v in wpb1 wpb6 wpb9; do
  echo -n "fhd $v  " >> $OUT/ab.txt
  FVVDP_PLACEMENT_PROBE=0 FVVDP_LIB=$R/build_variants/$v.so python bench.py $B --width 1920 --height 1080 --display standard_fhd 2>>$OUT/err_$v.txt | python -c "$P" >> $OUT/ab.txt 2>&1 || echo failed >> $OUT/ab.txt
done
cat $OUT/ab.txt
