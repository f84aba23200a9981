#!/usr/bin/env python3
"""YUV ingest timing probe: tools/gpu_yuv.py [HxWxN[:bits[:css]] ...]   (FVVDP_LIB / FVVDP_TEMPORAL_SCALAR honoured)"""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_yuv_pair
specs = sys.argv[1:] or ["2160x3840x60:8:420", "2160x3840x60:10:420"]
for sp in specs:
    parts = sp.split(":")
    H, W, N = (int(x) for x in parts[0].split("x"))
    bd = int(parts[1]) if len(parts) > 1 else 8
    css = parts[2] if len(parts) > 2 else "420"
    fps = int(parts[3]) if len(parts) > 3 else 30
    ty, ry = synth_yuv_pair(N, H, W, bit_depth=bd, chroma_ss=css, device="cuda")
    if os.environ.get("YUV_DARK"):          # luma just above black: every value on the sRGB toe (the slow branch of the display model)
        for a in (ty, ry):
            Y = a[:, :H * W]
            Y.copy_((16 * (1 << (bd - 8)) + (Y.to(torch.int64) - 16 * (1 << (bd - 8))) // 24).to(a.dtype))
    if bd > 8:
        ty, ry = ty.to(torch.int16), ry.to(torch.int16)
    m = fv.fvvdp(display_name=os.environ.get("PROBE_DISPLAY", "standard_4k"))      # PROBE_DISPLAY=standard_hdr_pq: the PQ display model
    vs = fv.fvvdp_video_source_yuv_frames(ty, ry, fps, W, H, bit_depth=bd, chroma_ss=css, color_space="bt709",
                                          display_photometry=m.display_photometry)
    m.timing = True
    best = 1e9
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q, st = m.predict_video_source(vs)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        if it == 1:
            ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
            nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    print("%s %s best %.2f ms %.0f Mpix/s JOD %.6f | temporal %.1f us/frame" % (
        os.path.basename(nat.LIB_PATH), sp, best * 1e3, 2 * W * H * N / best / 1e6, float(q), ms[0] / max(cnt[0], 1) / N * 1e3), flush=True)
