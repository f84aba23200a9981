#!/bin/bash
# round 4, session 41: band_kernel reading its arguments from the kernel-argument segment (no scalar spills in the map / view-map variants): speed check
R=$(pwd); OUT=$R/gpurun_out/r4s41; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu > $OUT/tests.txt 2>&1
tail -n 3 $OUT/tests.txt
rm -f $OUT/ab.txt
for rep in 1 2; do
  for v in base new; do
    L=""; [ $v = base ] && L="FVVDP_LIB=$R/build_variants/base.so"
    echo "== $v" >> $OUT/ab.txt
    env $L python tools/gpu_config4.py 2>/dev/null | grep -E "^kernel us/frame:" >> $OUT/ab.txt
    env $L python bench.py --no-cpu-baseline --no-h2d --no-measure-traffic 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], g["levels_us_per_frame_median"], g["us_per_frame_all_levels"])' >> $OUT/ab.txt
    env $L python tools/gpu_heatprof.py 12 threshold 2>/dev/null | grep "^total" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
