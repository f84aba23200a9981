#!/bin/bash
# round 4, session 12: spill-free fallback temporal kernels (tests, odd-size speed), cache policy of the coarse-level stores
R=$(pwd); OUT=$R/gpurun_out/r4s12; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
python - > $OUT/oddsize.txt 2>/dev/null <<'PY'
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
for (H, W) in ((768, 1366), (767, 1365), (1080, 1920), (1079, 1919)):
    for fps, N in ((30, 60), (60, 60), (120, 60)):
        t, r = synth_video_pair(N, H, W, device="cuda")
        m = fv.fvvdp(display_name="standard_fhd")
        for _ in range(3):
            q, _ = m.predict(t, r, frames_per_second=fps)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            q, _ = m.predict(t, r, frames_per_second=fps)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("%dx%d x%d @%d fps: %.3f ms per clip = %.1f Gpix/s  (%s temporal kernel)  JOD %.6f" % (
            W, H, N, fps, dt * 1e3, 2.0 * W * H * N / dt / 1e9, "vector" if (W * H) % 4 == 0 else "per-pixel fallback", float(q)), flush=True)
PY
cat $OUT/oddsize.txt
for rep in 1 2 3; do
  for v in default r4_staux0 r4_staux1 r4_staux3; do
    if [ $v = default ]; then L=""; else L="FVVDP_LIB=$R/build_variants/$v.so"; fi
    echo "== $v" >> $OUT/staux.txt
    env $L timeout 300 python tools/gpu_bandonly_speed.py 12 2>/dev/null | grep -v amdgpu >> $OUT/staux.txt
  done
done
cat $OUT/staux.txt
