#!/bin/bash
# round 4, session 4: level-0 scratch through the VMM API with physical chunks of different sizes against the default allocation
R=$(pwd); OUT=$R/gpurun_out/r4s4; mkdir -p $OUT
cd $R
B="--no-cpu-baseline --no-h2d --no-measure-traffic --steps 12 --warmup 4"
for rep in 1 2; do
  FVVDP_PIPELINE=0 FVVDP_PLACEMENT_PROBE=0 timeout 300 python bench.py $B > $OUT/default_$rep.json 2> $OUT/default_$rep.err
  for mb in 2 8 32 128 512; do
    FVVDP_PIPELINE=0 FVVDP_PLACEMENT_PROBE=0 FVVDP_ALLOC=vmm FVVDP_VMM_CHUNK_MB=$mb timeout 300 python bench.py $B > $OUT/vmm${mb}_$rep.json 2> $OUT/vmm${mb}_$rep.err
  done
  FVVDP_PIPELINE=0 FVVDP_PLACEMENT_PROBE=0 FVVDP_ALLOC=vmm FVVDP_VMM_CHUNK_MB=32 FVVDP_VMM_INTERLEAVE=4 timeout 300 python bench.py $B > $OUT/vmm32i4_$rep.json 2> $OUT/vmm32i4_$rep.err
  FVVDP_PIPELINE=0 FVVDP_PLACEMENT_PROBE=0 FVVDP_ALLOC_FLAGS=contiguous timeout 300 python bench.py $B > $OUT/contig_$rep.json 2> $OUT/contig_$rep.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"],"*.json"))):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    g=d.get("graded_pass",{})
    print("%-22s ms/pair %.3f  (K1 %.1f lvl01 %.1f pyr %.1f isolated)" % (os.path.basename(f), d["ms_per_pair"], g.get("temporal_us_per_frame_median",0), g["levels_us_per_frame_median"][0], g.get("us_per_frame_all_levels",0)))
PY
