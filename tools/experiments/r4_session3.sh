#!/bin/bash
# round 4, session 3: queued pairs (predict_batch) with and without the stage overlap; the revised placement selection
R=$(pwd); OUT=$R/gpurun_out/r4s3; mkdir -p $OUT
cd $R
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
for rep in 1 2 3; do
  for pipe in 0 2 3; do
    FVVDP_PIPELINE=$pipe FVVDP_PLACEMENT_PROBE=0 timeout 600 python bench.py $B --pairs-per-gpu 8 --steps 6 --warmup 2 > $OUT/q8_p${pipe}_$rep.json 2> $OUT/q8_p${pipe}_$rep.err
  done
  FVVDP_PIPELINE=2 FVVDP_PIPELINE_ORDER=natural FVVDP_PLACEMENT_PROBE=0 timeout 600 python bench.py $B --pairs-per-gpu 8 --steps 6 --warmup 2 > $OUT/q8_p2nat_$rep.json 2> $OUT/q8_p2nat_$rep.err
  for probe in 1 0; do
    FVVDP_PIPELINE=0 FVVDP_PLACEMENT_PROBE=$probe FVVDP_DEBUG_ALLOC=1 timeout 300 python bench.py $B --steps 20 --warmup 6 > $OUT/b_s${probe}_$rep.json 2> $OUT/b_s${probe}_$rep.err
  done
done
for rep in 4 5 6; do
  for probe in 1 0; do
    FVVDP_PIPELINE=0 FVVDP_PLACEMENT_PROBE=$probe FVVDP_DEBUG_ALLOC=1 timeout 300 python bench.py $B --steps 20 --warmup 6 > $OUT/b_s${probe}_$rep.json 2> $OUT/b_s${probe}_$rep.err
  done
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"],"*.json"))):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    g=d.get("graded_pass",{})
    print("%-22s ms/pair %.3f  (K1 %.1f pyr %.1f isolated) place %s %s" % (os.path.basename(f), d["ms_per_pair"], g.get("temporal_us_per_frame_median",0), g.get("us_per_frame_all_levels",0), d.get("placement",{}).get("us_per_frame_incumbent_candidate"), d.get("placement",{}).get("candidate_kept")))
PY
