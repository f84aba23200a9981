#!/usr/bin/env python3
"""Randomised sweep of the oracle against the REAL reference (build container only: imports /root/reference on torch-CPU, like
tools/gen_golden.py).  Same case generator as tools/experiments/gpu_stress.py (sizes, frame rates, paddings, dtypes, colour / gray,
displays, foveated), so that the chain reference -> oracle (here) -> HIP path (gpu_stress.py on the GPU box) is closed on the same
population of inputs.  Prints one summary line; the output is kept as profiles/rNN_oracle_vs_reference.txt.
usage: tools/oracle_vs_reference.py [cases] [seed]          (RGB / gray arrays, generator of gpu_stress.py)
       tools/oracle_vs_reference.py yuv [cases] [seed]      (raw planar YUV through the reference's video_reader_yuv_pytorch.unpack,
                                                             generator of gpu_stress_yuv.py)
       tools/oracle_vs_reference.py shapes                  (the extreme frame shapes of gpu_stress_shapes.py, image and 5-frame video)
       tools/oracle_vs_reference.py heat [cases] [seed]     ('raw' difference maps, generator of gpu_stress_heat.py)
       tools/oracle_vs_reference.py resize [cases] [seed]   (unpack with resize_fn: the full-screen resize, random sizes and formats)"""
import os, sys, types, logging
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fvvdp_oracle as orc
from fovvideovdp_amd.synth import synth_video_pair


def import_reference():
    for name in ("imageio", "imageio.v2", "ffmpeg"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["imageio"].v2 = sys.modules["imageio.v2"]
    sys.path.insert(0, os.environ.get("FVVDP_REFERENCE", "/root/reference"))
    import pyfvvdp
    return pyfvvdp


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    logging.disable(logging.WARNING)
    pyfvvdp = import_reference()
    torch.set_num_threads(8)
    # frame size range as in gpu_stress.py (HLO/HHI/WLO/WHI/NMAX in the environment for mid-size sweeps)
    HLO, HHI = int(os.environ.get("HLO", 17)), int(os.environ.get("HHI", 150))
    WLO, WHI = int(os.environ.get("WLO", 17)), int(os.environ.get("WHI", 260))
    NMAX = int(os.environ.get("NMAX", 14))
    worst, worst_q, fails, both_raise = (0.0, None), (0.0, None), 0, 0
    for case in range(n_cases):
        H, W = int(rng.integers(HLO, HHI)), int(rng.integers(WLO, WHI))
        fps = int(rng.choice([0, 24, 25, 30, 50, 60, 120, 144, 240]))
        N = 1 if fps == 0 else int(rng.integers(2, NMAX))
        pad = str(rng.choice(["replicate", "circular", "pingpong"]))
        dt = str(rng.choice(["u8", "u16", "f32"]))
        C_ch = int(rng.choice([1, 3]))
        disp = str(rng.choice(["standard_4k", "standard_fhd", "standard_hdr_pq", "standard_hmd"]))
        fov = bool(rng.integers(0, 4) == 0)
        t, r = synth_video_pair(N, H, W, pair=int(rng.integers(0, 50)))
        if C_ch == 1:
            t, r = t[:, 1:2], r[:, 1:2]
        tn, rn = t.numpy(), r.numpy()
        if dt == "u16":
            tn, rn = tn.astype(np.uint16) * 257, rn.astype(np.uint16) * 257
        elif dt == "f32":
            tn, rn = tn.astype(np.float32) / np.float32(255), rn.astype(np.float32) / np.float32(255)
        desc = f"{W}x{H}x{N} fps={fps} pad={pad} {dt} C={C_ch} {disp} fov={fov}"
        fix = np.array([W * 0.3, H * 0.6]) if fov else None
        rq = oq = None
        try:
            m = pyfvvdp.fvvdp(display_name=disp, temp_padding=pad, foveated=fov, heatmap=None, device=torch.device("cpu"), quiet=True)
            with torch.no_grad():                # numpy uint16 goes through the reference's own int16 packing (video_source.py:129-137)
                rq, rst = m.predict(tn, rn, dim_order="BCFHW", frames_per_second=fps,
                                    fixation_point=None if fix is None else torch.tensor(fix, dtype=torch.float32))
        except Exception as e:
            rerr = str(e)[:70]
        try:
            oq, ost = orc.Oracle(disp, temp_padding=pad, foveated=fov).predict(tn, rn, "BCFHW", fps, fix)
        except Exception as e:
            oerr = str(e)[:70]
        if rq is None and oq is None:
            both_raise += 1
            continue
        if rq is None or oq is None:
            print("FAIL (one side raised)", desc, "| reference:", "ok" if rq is not None else rerr, "| oracle:", "ok" if oq is not None else oerr)
            fails += 1
            continue
        dq = abs(float(rq) - float(oq))
        a = np.asarray(ost["Q_per_ch"], np.float64)
        b = np.asarray(rst["Q_per_ch"], np.float64)
        rel = float(np.max(np.abs(a - b) / (np.abs(b) + 1e-5 * b.max() + 1e-12))) if a.shape == b.shape else 1.0
        if dq > worst[0]: worst = (dq, desc)
        if rel > worst_q[0]: worst_q = (rel, desc)
        if dq > 5e-4 or rel > 2e-2:
            print("FAIL", desc, "dJOD %.2e relQ %.2e" % (dq, rel))
            fails += 1
    print("cases", n_cases, "seed", seed, "both raise", both_raise, "fails", fails,
          "| worst dJOD %.2e (%s) | worst rel Q %.2e (%s)" % (worst[0], worst[1], worst_q[0], worst_q[1]))


def main_yuv():
    from fovvideovdp_amd.synth import synth_yuv_pair
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(seed)
    logging.disable(logging.WARNING)
    pyfvvdp = import_reference()
    import pyfvvdp.video_source_file as vsf
    from pyfvvdp.video_source import fvvdp_video_source_dm, reshuffle_dims
    torch.set_num_threads(8)
    worst, fails = (0.0, None), 0
    for case in range(n_cases):
        css = str(rng.choice(["420", "444"]))
        H, W = int(rng.integers(9, 70)) * 2, int(rng.integers(9, 120)) * 2
        bd = int(rng.choice([8, 10, 12]))
        fps = int(rng.choice([24, 30, 50, 60, 120]))
        N = int(rng.integers(2, 12))
        cs = str(rng.choice(["bt709", "bt2020nc"]))
        disp = str(rng.choice(["standard_4k", "standard_fhd", "standard_hdr_pq"]))
        ty, ry = synth_yuv_pair(N, H, W, bit_depth=bd, chroma_ss=css, pair=int(rng.integers(0, 9)))
        desc = f"{W}x{H}x{N} {css} {bd}bit {cs} fps={fps} {disp}"
        tn = ty.numpy() if bd == 8 else ty.numpy().astype(np.uint16)
        rn = ry.numpy() if bd == 8 else ry.numpy().astype(np.uint16)
        # the reference's reader object without ffmpeg (as tools/gen_golden.py g7): only unpack() is used (video_source_file.py:219-276)
        rd = object.__new__(vsf.video_reader_yuv_pytorch)
        rd.width, rd.height, rd.bit_depth, rd.chroma_ss, rd.color_space = W, H, bd, css, cs
        rd.y_pixels, rd.y_shape = W * H, (H, W)
        rd.uv_shape = (H // 2, W // 2) if css == "420" else (H, W)
        rd.uv_pixels = rd.uv_shape[0] * rd.uv_shape[1]
        csn = "BT.2020" if cs == "bt2020nc" else "sRGB"

        class Src(fvvdp_video_source_dm):
            def __init__(self):
                super().__init__(display_photometry=disp, color_space_name=csn)

            def get_video_size(self):
                return (H, W, N)

            def get_frames_per_second(self):
                return fps

            def _fr(self, arr, f, device):
                t = reshuffle_dims(rd.unpack(arr[f], device), in_dims='HWC', out_dims="BCFHW")
                L = self.dm_photometry.forward(t)
                c = self.color_to_luminance
                return L[:, 0:1] * c[0] + L[:, 1:2] * c[1] + L[:, 2:3] * c[2]

            def get_test_frame(self, f, device):
                return self._fr(tn, f, device)

            def get_reference_frame(self, f, device):
                return self._fr(rn, f, device)

        m = pyfvvdp.fvvdp(display_name=disp, heatmap=None, device=torch.device("cpu"), quiet=True)
        with torch.no_grad():
            rq, rst = m.predict_video_source(Src())
        oq, ost = orc.Oracle(disp, color_space=csn).predict_yuv(tn, rn, fps, W, H, bit_depth=bd, chroma_ss=css, color_space=cs)
        dq = abs(float(rq) - float(oq))
        if dq > worst[0]: worst = (dq, desc)
        if dq > 5e-4:
            print("FAIL", desc, "dJOD %.2e" % dq)
            fails += 1
    print("yuv cases", n_cases, "seed", seed, "fails", fails, "| worst dJOD %.2e (%s)" % worst)


def main_shapes():
    logging.disable(logging.WARNING)
    pyfvvdp = import_reference()
    torch.set_num_threads(8)
    shapes = [(16, 4096), (17, 3001), (2048, 16), (1999, 17), (16, 16), (16, 17), (17, 16), (31, 33), (64, 120), (64, 121), (64, 124),
              (64, 125), (64, 240), (64, 241), (64, 244), (64, 245), (33, 247), (33, 248), (33, 249), (33, 256), (33, 257), (120, 64),
              (121, 63), (255, 255), (256, 256), (257, 257), (40, 1024), (41, 1023), (300, 500)]
    worst, worst_q, fails, both = (0.0, None), (0.0, None), 0, 0
    for (H, W) in shapes:
        for (N, fps) in ((1, 0), (5, 30)):
            t, r = synth_video_pair(N, H, W, pair=3)
            tn, rn = t.numpy(), r.numpy()
            desc = f"{W}x{H}x{N}"
            rq = oq = None
            try:
                m = pyfvvdp.fvvdp(display_name="standard_4k", heatmap=None, device=torch.device("cpu"), quiet=True)
                with torch.no_grad():
                    rq, rst = m.predict(tn, rn, dim_order="BCFHW", frames_per_second=fps)
            except Exception as e:
                rerr = str(e)[:60]
            try:
                oq, ost = orc.Oracle("standard_4k").predict(tn, rn, frames_per_second=fps)
            except Exception as e:
                oerr = str(e)[:60]
            if rq is None and oq is None:
                both += 1
                continue
            if rq is None or oq is None:
                print("one side raised:", desc, "| reference:", "ok" if rq is not None else rerr, "| oracle:", "ok" if oq is not None else oerr)
                fails += 1
                continue
            dq = abs(float(rq) - float(oq))
            a, b = np.asarray(ost["Q_per_ch"], np.float64), np.asarray(rst["Q_per_ch"], np.float64)
            rel = float(np.max(np.abs(a - b) / (np.abs(b) + 1e-5 * b.max() + 1e-12))) if a.shape == b.shape else 1.0
            if dq > worst[0]: worst = (dq, desc)
            if rel > worst_q[0]: worst_q = (rel, desc)
            if dq > 5e-4 or rel > 1e-2:
                print("FAIL", desc, "dJOD %.2e relQ %.2e" % (dq, rel))
                fails += 1
    print("shapes", 2 * len(shapes), "both raise", both, "fails", fails,
          "| worst dJOD %.2e (%s) | worst rel Q %.2e (%s)" % (worst[0], worst[1], worst_q[0], worst_q[1]))


def main_heat():
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(seed)
    logging.disable(logging.WARNING)
    pyfvvdp = import_reference()
    torch.set_num_threads(8)
    worst, worst_m, fails = (0.0, None), (0.0, None), 0
    for case in range(n_cases):
        H, W = int(rng.integers(17, 150)), int(rng.integers(17, 260))
        fps = int(rng.choice([0, 24, 30, 60, 120]))
        N = 1 if fps == 0 else int(rng.integers(2, 9))
        pad = str(rng.choice(["replicate", "circular", "pingpong"]))
        disp = str(rng.choice(["standard_4k", "standard_fhd", "standard_hdr_pq", "standard_hmd"]))
        fov = bool(rng.integers(0, 3) == 0)
        t, r = synth_video_pair(N, H, W, pair=int(rng.integers(0, 50)))
        fix = np.array([W * 0.3, H * 0.6]) if fov else None
        desc = f"{W}x{H}x{N} fps={fps} pad={pad} {disp} fov={fov}"
        try:
            m = pyfvvdp.fvvdp(display_name=disp, temp_padding=pad, foveated=fov, heatmap="raw", device=torch.device("cpu"), quiet=True)
            with torch.no_grad():
                rq, rst = m.predict(t, r, dim_order="BCFHW", frames_per_second=fps,
                                    fixation_point=None if fix is None else torch.tensor(fix, dtype=torch.float32))
            oq, ost = orc.Oracle(disp, temp_padding=pad, foveated=fov, heatmap="raw").predict(t.numpy(), r.numpy(), "BCFHW", fps, fix)
        except Exception as e:
            print("raise", desc, str(e)[:80])
            continue
        g, h = rst["heatmap"].float().numpy().astype(np.float64), ost["heatmap"].astype(np.float64)
        rel = float(np.max(np.abs(h - g) / np.maximum(np.abs(g), 2e-3))) * 1024
        mean = float(np.mean(np.abs(h - g)))
        if rel > worst[0]: worst = (rel, desc)
        if mean > worst_m[0]: worst_m = (mean, desc)
        if rel > 3.0 or abs(float(rq) - float(oq)) > 5e-5:
            print("FAIL", desc, "ulp %.2f dJOD %.2e" % (rel, abs(float(rq) - float(oq))))
            fails += 1
    print("heat cases", n_cases, "seed", seed, "fails", fails,
          "| worst %.2f fp16 ulp (%s) | worst mean abs %.2e (%s)" % (worst[0], worst[1], worst_m[0], worst_m[1]))


def main_resize():
    """The full-screen resize: oracle.yuv_unpack with resize_fn against the reference's video_reader_yuv_pytorch.unpack with resize_fn
    (video_source_file.py:238-244) on random source / target sizes, bit depths, subsamplings and colour matrices."""
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 6617
    import_reference()
    import pyfvvdp.video_source_file as vsf
    from fovvideovdp_amd.synth import synth_yuv_pair
    rng = np.random.default_rng(seed)
    worst = {}
    for k in range(n_cases):
        fn = ("bilinear", "bicubic", "nearest", "area")[k % 4]
        bd, css = ((8, "420"), (10, "444"), (12, "420"), (16, "444"))[rng.integers(0, 4)]
        cs = ("bt709", "bt2020nc")[rng.integers(0, 2)]
        H, W = 2 * int(rng.integers(5, 40)), 2 * int(rng.integers(5, 60))
        Ho, Wo = int(rng.integers(7, 130)), int(rng.integers(7, 170))
        ty, _ = synth_yuv_pair(2, H, W, bit_depth=bd, chroma_ss=css)
        a = ty.numpy() if bd == 8 else ty.numpy().astype(np.uint16)
        rd = object.__new__(vsf.video_reader_yuv_pytorch)
        rd.width, rd.height, rd.bit_depth, rd.chroma_ss, rd.color_space = W, H, bd, css, cs
        rd.y_pixels, rd.y_shape = W * H, (H, W)
        rd.uv_shape = (H // 2, W // 2) if css == "420" else (H, W)
        rd.uv_pixels = rd.uv_shape[0] * rd.uv_shape[1]
        rd.resize_fn, rd.resize_height, rd.resize_width = fn, Ho, Wo
        ref = rd.unpack(a[1], torch.device("cpu")).numpy()
        o = orc.yuv_unpack(a[1], W, H, bd, css, cs, fn, (Ho, Wo))
        worst[fn] = max(worst.get(fn, 0.0), float(np.abs(ref - o).max()))
    fails = sum(v > 8e-6 for v in worst.values())
    print("resize cases %d seed %d fails %d | worst |dRGB| per method: %s" % (
        n_cases, seed, fails, ", ".join("%s %.2e" % kv for kv in sorted(worst.items()))))


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else ""
    (main_yuv() if mode == "yuv" else main_shapes() if mode == "shapes" else main_heat() if mode == "heat" else
     main_resize() if mode == "resize" else main())
