#!/bin/bash
# round 4, session 6: clamp-free variant of the two-level kernel: parity tests + A/B
R=$(pwd); OUT=$R/gpurun_out/r4s6; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
B="--no-cpu-baseline --no-h2d --no-measure-traffic --steps 20 --warmup 5"
for rep in 1 2 3; do
  timeout 300 python bench.py $B > $OUT/b_inrange_$rep.json 2> $OUT/b_inrange_$rep.err
  FVVDP_BAND_INRANGE=0 timeout 300 python bench.py $B > $OUT/b_clamps_$rep.json 2> $OUT/b_clamps_$rep.err
done
timeout 300 python bench.py $B --width 1920 --height 1080 --display standard_fhd > $OUT/fhd_inrange.json 2> $OUT/fhd_inrange.err
FVVDP_BAND_INRANGE=0 timeout 300 python bench.py $B --width 1920 --height 1080 --display standard_fhd > $OUT/fhd_clamps.json 2> $OUT/fhd_clamps.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"],"*.json"))):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    g=d.get("graded_pass",{})
    print("%-22s ms/pair %.3f  (K1 %.1f lvl01 %.2f pyr %.2f isolated) jod %s" % (os.path.basename(f), d["ms_per_pair"], g.get("temporal_us_per_frame_median",0), g["levels_us_per_frame_median"][0], g.get("us_per_frame_all_levels",0), d["jod"][:2]))
PY
