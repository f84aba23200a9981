#!/bin/bash
# round 4, session 5: the whole GPU suite on the chunk-mapped scratch, then the bench line with both allocations
R=$(pwd); OUT=$R/gpurun_out/r4s5; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
B="--no-cpu-baseline --no-h2d --no-measure-traffic --steps 20 --warmup 5"
for rep in 1 2 3; do
  timeout 300 python bench.py $B > $OUT/b_vmm_$rep.json 2> $OUT/b_vmm_$rep.err
  FVVDP_ALLOC=malloc timeout 300 python bench.py $B > $OUT/b_malloc_$rep.json 2> $OUT/b_malloc_$rep.err
done
timeout 600 python bench.py $B --pairs-per-gpu 8 --steps 6 --warmup 2 > $OUT/q8_vmm.json 2> $OUT/q8_vmm.err
FVVDP_ALLOC=malloc timeout 600 python bench.py $B --pairs-per-gpu 8 --steps 6 --warmup 2 > $OUT/q8_malloc.json 2> $OUT/q8_malloc.err
timeout 300 python bench.py $B --width 1920 --height 1080 --display standard_fhd > $OUT/fhd_vmm.json 2> $OUT/fhd_vmm.err
FVVDP_ALLOC=malloc timeout 300 python bench.py $B --width 1920 --height 1080 --display standard_fhd > $OUT/fhd_malloc.json 2> $OUT/fhd_malloc.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"],"*.json"))):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    g=d.get("graded_pass",{})
    print("%-22s ms/pair %.3f  (K1 %.1f lvl01 %.1f pyr %.1f isolated) jod %s" % (os.path.basename(f), d["ms_per_pair"], g.get("temporal_us_per_frame_median",0), g["levels_us_per_frame_median"][0], g.get("us_per_frame_all_levels",0), d["jod"][:2]))
PY
# context creation cost
python - <<'PY'
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
t, r = synth_video_pair(60, 2160, 3840, device="cuda")
for rep in range(2):
    m = fv.fvvdp(display_name="standard_4k")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.predict(t, r, frames_per_second=30)
    t1 = time.perf_counter()
    m.predict(t, r, frames_per_second=30)
    t2 = time.perf_counter()
    print("first call %.1f ms (context creation + first touch), second %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    del m
PY
