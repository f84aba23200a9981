#!/usr/bin/env python3
"""Output of the temporal kernels (pyramid level 0: the four temporal channels) for closed-form sources, to compare two builds of the library
(FVVDP_LIB=...): a SHA-1 of the exported level per case, and with `save DIR` / `cmp DIR` the largest difference between the two builds relative to
the luminance behind the pixel.  Round 6: the packed (test, reference) display model of the 16-bit / float temporal kernels.
usage: tools/experiments/gpu_k1_bits.py [save DIR | cmp DIR]   (one line per case)"""
import ctypes as C, hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W, N = 216, 384, 20
MODE, DIR = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else (None, None)
if DIR:
    os.makedirs(DIR, exist_ok=True)
for disp in ("standard_4k", "standard_hdr_pq", "standard_hdr_linear", "sdr_fhd_24"):
    for kind in ("u16", "f32rgb", "f32gray", "u16gray", "f32oob"):
        for fps in (30, 60, 120, 240):
            test, ref = synth_video_pair(N, H, W, device="cuda")
            if kind.startswith("u16"):
                test, ref = (test.to(torch.int32) * 257 + 3).clamp(0, 65535).to(torch.int16), (ref.to(torch.int32) * 257).to(torch.int16)
                if kind == "u16gray":
                    test, ref = test[:, 1:2].contiguous(), ref[:, 1:2].contiguous()
            else:
                s = 900.0 if "linear" in disp else 1.0
                test, ref = test.float() / 255 * s, ref.float() / 255 * s
                if kind == "f32gray":
                    test, ref = test[:, 1:2].contiguous(), ref[:, 1:2].contiguous()
                if kind == "f32oob":
                    test = test * 1.6 - 0.3            # the synthetic clip spans 0.14 .. 0.89: samples outside [0,1] (flag + clip)
            m = fv.fvvdp(display_name=disp, quiet=True)
            try:
                q, st = m.predict(test, ref, frames_per_second=fps)
            except Exception as e:
                print(disp, kind, fps, "raised", type(e).__name__)
                continue
            out = torch.empty((N, 4, H, W), dtype=torch.float32, device="cuda")
            nat.check(nat.lib().fvvdp_export_level(m._ctx.handle, 0, N, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            arr = out.cpu().numpy()
            h = hashlib.sha1(arr.tobytes()).hexdigest()[:16]
            extra = ""
            fn = os.path.join(DIR, "%s_%s_%d.npy" % (disp, kind, fps)) if DIR else None
            if MODE == "save":
                np.save(fn, arr)
            elif MODE == "cmp":
                o = np.load(fn)
                ndiff = int((o != arr).sum())
                scale = np.maximum(np.abs(o[:, :2]), 1e-3)            # the luminance behind a pixel (sustained channel), as the tests scale it
                rel = np.abs(o - arr) / np.concatenate([scale, scale], axis=1)
                k = np.unravel_index(int(np.argmax(rel)), rel.shape)
                extra = " | differ %d of %d, max |d| / luminance %.3g at %s (old %.8g new %.8g)" % (
                    ndiff, o.size, float(rel[k]), tuple(int(x) for x in k), float(o[k]), float(arr[k]))
            print(disp, kind, fps, "level0", h, "JOD", float(q).hex(), "finite", bool(np.isfinite(arr).all()), extra, flush=True)
