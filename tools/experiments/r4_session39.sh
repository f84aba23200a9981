#!/bin/bash
# round 4, session 39: soft cross-workgroup sync of the strip groups of one (frame, chunk) row every N stages (experiment build)
R=$(pwd); OUT=$R/gpurun_out/r4s39; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0 FVVDP_LIB=$R/build_variants/rowsync.so FVVDP_BAND2_SPIN=300
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"], "JOD", d.get("jod"))'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
run() { # label, env, bench args
  echo -n "$1  " >> $OUT/scan.txt
  env $2 python bench.py $B $3 2>>$OUT/err.txt | python -c "$P" >> $OUT/scan.txt 2>&1 || echo failed >> $OUT/scan.txt
}
for rep in 1 2; do
  for ev in 0 1 2 4 8 16 40; do
    run "4k60 sync_every=$ev" "FVVDP_BAND2_SYNC=$ev" ""
  done
done
cat $OUT/scan.txt
FVVDP_BAND2_SYNC=4 FVVDP_LIB=$R/build_variants/rowsync_tl.so python tools/gpu_timeline.py > $OUT/timeline_sync4.txt 2>>$OUT/err.txt
cat $OUT/timeline_sync4.txt
