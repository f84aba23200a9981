#!/bin/bash
# round 4, session 20: the waves of adjacent strips kept in step (BAND2_WPB waves per workgroup, one barrier per stage)
R=$(pwd); OUT=$R/gpurun_out/r4s20; mkdir -p $OUT
cd $R
FVVDP_LIB=$R/build_variants/wpb4.so timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "two_level or clamp_free" > $OUT/tests_wpb4.txt 2>&1
tail -n 3 $OUT/tests_wpb4.txt
FVVDP_LIB=$R/build_variants/wpb2.so timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "two_level and not foveated" > $OUT/tests_wpb2.txt 2>&1
tail -n 3 $OUT/tests_wpb2.txt
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"], "JOD", d.get("jod"))'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/ab.txt
for rep in 1 2 3; do
  for v in wpb1 wpb2 wpb4; do
    echo -n "$v  " >> $OUT/ab.txt
    FVVDP_PLACEMENT_PROBE=0 FVVDP_LIB=$R/build_variants/$v.so python bench.py $B 2>$OUT/err_$v.txt | python -c "$P" >> $OUT/ab.txt 2>&1 || echo failed >> $OUT/ab.txt
  done
done
for v in wpb1 wpb2 wpb4; do
  echo -n "fhd $v  " >> $OUT/ab.txt
  FVVDP_PLACEMENT_PROBE=0 FVVDP_LIB=$R/build_variants/$v.so python bench.py $B --width 1920 --height 1080 --display standard_fhd 2>>$OUT/err_$v.txt | python -c "$P" >> $OUT/ab.txt 2>&1 || echo failed >> $OUT/ab.txt
done
cat $OUT/ab.txt
FVVDP_LIB=$R/build_variants/timeline4.so python tools/gpu_timeline.py > $OUT/timeline_4k_wpb4.txt 2>>$OUT/err.txt
cat $OUT/timeline_4k_wpb4.txt
