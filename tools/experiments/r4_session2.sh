#!/bin/bash
# round 4, session 2: what makes the temporal kernel and the pyramid pass co-run well?  order, priorities, CU partitions
R=$(pwd); OUT=$R/gpurun_out/r4s2; mkdir -p $OUT
cd $R
B="--no-cpu-baseline --no-h2d --no-measure-traffic --steps 20 --warmup 6"
run() { # name, env...
  name=$1; shift
  env "$@" FVVDP_PLACEMENT_PROBE=0 timeout 300 python bench.py $B > $OUT/$name.json 2> $OUT/$name.err
  env "$@" FVVDP_PLACEMENT_PROBE=0 timeout 600 python bench.py $B --pairs-per-gpu 4 --steps 6 --warmup 2 > $OUT/${name}_q4.json 2>> $OUT/$name.err
}
for rep in 1 2; do
run seq_$rep FVVDP_PIPELINE=0
run p2_$rep FVVDP_PIPELINE=2
run p2_k2first_$rep FVVDP_PIPELINE=2 FVVDP_PIPELINE_ORDER=k2first
run p2_prio_k2_$rep FVVDP_PIPELINE=2 FVVDP_PIPELINE_PRIO=k2
run p2_prio_k1_$rep FVVDP_PIPELINE=2 FVVDP_PIPELINE_PRIO=k1
for k in 6 8 10 12 16; do
  run p2_cu${k}_$rep FVVDP_PIPELINE=2 FVVDP_PIPELINE_CUS=$k
  run p2_cu${k}_all_$rep FVVDP_PIPELINE=2 FVVDP_PIPELINE_CUS=$k FVVDP_PIPELINE_K2ALL=1
done
run p4_cu8_$rep FVVDP_PIPELINE=4 FVVDP_PIPELINE_CUS=8
run p4_cu12_$rep FVVDP_PIPELINE=4 FVVDP_PIPELINE_CUS=12
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ["OUT"],"*.json"))):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f),"FAILED",e); continue
    g=d.get("graded_pass",{})
    print("%-28s ms/pair %.3f  (K1 %.1f pyr %.1f isolated)" % (os.path.basename(f), d["ms_per_pair"], g.get("temporal_us_per_frame_median",0), g.get("us_per_frame_all_levels",0)))
PY
