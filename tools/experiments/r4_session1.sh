#!/bin/bash
# round 4, session 1: GPU tests with the stage overlap + sum-based placement selection, then A/B of the bench line
R=$(pwd); OUT=$R/gpurun_out/r4s1; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_config5_per_gpu_share_8_pairs_golden > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
B="--no-cpu-baseline --no-h2d --no-measure-traffic --steps 20 --warmup 6"
for rep in 1 2 3; do
  for pipe in 0 2 3 4; do
    for probe in 1 0; do
      FVVDP_PIPELINE=$pipe FVVDP_PLACEMENT_PROBE=$probe FVVDP_DEBUG_ALLOC=1 timeout 300 python bench.py $B > $OUT/b_p${pipe}_s${probe}_$rep.json 2> $OUT/b_p${pipe}_s${probe}_$rep.err
    done
  done
done
for pipe in 0 2 3; do
  FVVDP_PIPELINE=$pipe timeout 600 python bench.py $B --pairs-per-gpu 8 --steps 5 --warmup 2 > $OUT/b8_p${pipe}.json 2> $OUT/b8_p${pipe}.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("OUT","gpurun_out/r4s1"),"b*.json"))):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    g=d.get("graded_pass",{})
    print(os.path.basename(f), "ms/step", d["ms_per_step"], "mean", d["timing"]["ms_per_step_mean"], "K1", g.get("temporal_us_per_frame_median"), "pyr", g.get("us_per_frame_all_levels"), "place", d.get("placement",{}).get("us_per_frame_incumbent_candidate"), d.get("placement",{}).get("candidate_kept"))
PY
