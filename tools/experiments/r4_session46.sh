#!/bin/bash
# round 4, session 46: occupancy of the uint8 temporal kernel capped through unused dynamic LDS (16 / 12 / 10 / 8 / 6 workgroups per CU)
# (needs the experiment patch of that session: launch_vec passing FVVDP_K1_LDS_PAD bytes of dynamic LDS, -DK1_LDS_PAD_EXPERIMENT; not in the tree)
R=$(pwd); OUT=$R/gpurun_out/r4s46; mkdir -p $OUT
cd $R
export FVVDP_PLACEMENT_PROBE=0 FVVDP_LIB=$R/build_variants/k1pad.so
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); g=d["graded_pass"]; print(d["ms_per_step"], "K1", g["temporal_us_per_frame_median"], "lv01", g["levels_us_per_frame_median"][0], "all", g["us_per_frame_all_levels"])'
B="--no-cpu-baseline --no-h2d --no-measure-traffic"
rm -f $OUT/scan.txt
# one process per setting would draw a new allocation each time; the K1-only probe keeps ONE context and changes the pad between calls
python - <<'PY' > $OUT/pads.txt 2>$OUT/err.txt
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
import torch, fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
t, r = synth_video_pair(60, 2160, 3840, device="cuda")
ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
for ctxno in range(3):
    m = fv.fvvdp(display_name="standard_4k"); m.timing = True
    for _ in range(2): m.predict(t, r, frames_per_second=30)
    for rnd in range(2):
        row = []
        for pad, label in ((0, "16/CU"), (12288, "8/CU"), (18432, "6/CU"), (24576, "5/CU"), (32768, "4/CU"), (45056, "3/CU"), (57344, "2/CU")):
            os.environ["FVVDP_K1_LDS_PAD"] = str(pad)
            v = []
            for rep in range(3):
                m.predict(t, r, frames_per_second=30)
                nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
                v.append(ms[0] / 60 * 1e3)
            row.append("%s %.2f" % (label, float(np.median(v))))
        print("context %d round %d  K1 us per frame: %s" % (ctxno, rnd, " | ".join(row)))
PY
cat $OUT/pads.txt; tail -n 2 $OUT/err.txt
