#!/bin/bash
# round 4, session 44: cache policy of the temporal kernel's level-0 stores (aux bits of buffer_store: 1 sc0, 2 nt, 16 sc1)
R=$(pwd); OUT=$R/gpurun_out/r4s44; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"], "JOD", d.get("jod"))'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
for rep in 1 2; do
  for a in 2 0 1 3 16 18; do
    echo -n "aux=$a  " >> $OUT/scan.txt
    FVVDP_LIB=$R/build_variants/k1aux$a.so python bench.py $B 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
  done
  for a in 2 0 16; do
    echo -n "malloc aux=$a  " >> $OUT/scan.txt
    FVVDP_ALLOC=malloc FVVDP_LIB=$R/build_variants/k1aux$a.so python bench.py $B 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
  done
done
cat $OUT/scan.txt
