"""Portable synthetic video pairs for parity tests and the benchmark.

Integer-only, counter-based generator written with torch integer ops so that the very same code produces
bit-identical uint8 frames on the CPU (golden generation, oracle runs) and on the GPU (bench, parity tests):

  ref  = clip(128 + 64*sin(2pi(x/97 + f/31))*cos(2pi y/61) + (h1 mod 33 - 16) + 8*(c-1))
  test = clip(ref + (h2 mod 11 - 5) + 6*[f mod 7 == 0])

with h = lowbias32 hash of (seed, f, c, y, x) and sin/cos from a 1024-entry integer table.  Spatial noise plus
a periodic flicker excite every spatial band and both temporal channels.
"""
import math

import torch

_TAB_BITS = 10
_TAB = [int(round(math.sin(2.0 * math.pi * i / (1 << _TAB_BITS)) * 1024.0)) for i in range(1 << _TAB_BITS)]


def _hash32(x):
    m = 0xFFFFFFFF
    x = x & m
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & m
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & m
    x = x ^ (x >> 16)
    return x


def synth_frame_pair(f, H, W, C=3, seed_ref=1234, seed_test=5678, device="cpu"):
    """One (test, ref) frame pair, uint8 [C,H,W]."""
    dev = torch.device(device)
    tab = torch.tensor(_TAB, dtype=torch.int64, device=dev)
    n = 1 << _TAB_BITS
    x = torch.arange(W, dtype=torch.int64, device=dev).view(1, 1, W)
    y = torch.arange(H, dtype=torch.int64, device=dev).view(1, H, 1)
    c = torch.arange(C, dtype=torch.int64, device=dev).view(C, 1, 1)
    px = (torch.div(x * n, 97, rounding_mode="floor") + (f * n) // 31) % n
    py = torch.div(y * n, 61, rounding_mode="floor") % n
    s = tab[px]
    co = tab[(py + n // 4) % n]
    wave = torch.div(64 * s * co, 1 << 20, rounding_mode="floor")
    lin = ((f * C + c) * H + y) * W + x
    h1 = _hash32(lin + (seed_ref << 8) * 0x9E3779B1)
    h2 = _hash32(lin + (seed_test << 8) * 0x9E3779B1)
    ref = torch.clamp(128 + wave + (h1 % 33 - 16) + 8 * (c - 1), 0, 255)
    flick = 6 if (f % 7) == 0 else 0
    test = torch.clamp(ref + (h2 % 11 - 5) + flick, 0, 255)
    return test.to(torch.uint8), ref.to(torch.uint8)


def synth_video_pair(N, H, W, C=3, seed_ref=1234, seed_test=5678, device="cpu", pair=0):
    """Test/reference videos, uint8 BCFHW = [1,C,N,H,W].  `pair` offsets the seeds (seed + 1000*pair)."""
    dev = torch.device(device)
    test = torch.empty((1, C, N, H, W), dtype=torch.uint8, device=dev)
    ref = torch.empty((1, C, N, H, W), dtype=torch.uint8, device=dev)
    for f in range(N):
        t, r = synth_frame_pair(f, H, W, C, seed_ref + 1000 * pair, seed_test + 1000 * pair, dev)
        test[0, :, f] = t
        ref[0, :, f] = r
    return test, ref


def synth_gaze(N, H, W):
    """Gaze moving linearly from the top-left to the bottom-right corner, fp32 [N,2] as (x, y)
    (same path as pytorch_examples/ex_foveated_video.py:36-37 of the reference)."""
    gx = torch.linspace(0, W - 1, N, dtype=torch.float64)
    gy = torch.linspace(0, H - 1, N, dtype=torch.float64)
    return torch.stack((gx, gy), dim=1).to(torch.float32)


def synth_yuv_pair(N, H, W, bit_depth=8, chroma_ss="420", device="cpu", pair=0):
    """Raw planar limited-range YUV frames (Y plane, then U, then V per frame; [N, frame_elems]) derived from the
    synthetic RGB pair with integer arithmetic only (bit-identical on CPU and GPU).  uint8 for 8 bit, int32-exact
    values stored as uint16 (torch.int16 carrier is avoided: returned as int32 tensors to be cast by the caller)."""
    test, ref = synth_video_pair(N, H, W, device=device, pair=pair)
    out = []
    for v in (test, ref):
        rgb = v[0].permute(1, 0, 2, 3).to(torch.int64)                  # [N,3,H,W]
        R, G, B = rgb[:, 0], rgb[:, 1], rgb[:, 2]
        Y = torch.div(66 * R + 129 * G + 25 * B + 128, 256, rounding_mode="floor") + 16
        U = torch.div(-38 * R - 74 * G + 112 * B + 128, 256, rounding_mode="floor") + 128
        V = torch.div(112 * R - 94 * G - 18 * B + 128, 256, rounding_mode="floor") + 128
        if chroma_ss == "420":
            U = torch.div(U[:, 0::2, 0::2] + U[:, 0::2, 1::2] + U[:, 1::2, 0::2] + U[:, 1::2, 1::2] + 2, 4, rounding_mode="floor")
            V = torch.div(V[:, 0::2, 0::2] + V[:, 0::2, 1::2] + V[:, 1::2, 0::2] + V[:, 1::2, 1::2] + 2, 4, rounding_mode="floor")
        sc = 1 << (bit_depth - 8)
        fr = torch.cat([(Y * sc).reshape(N, -1), (U * sc).reshape(N, -1), (V * sc).reshape(N, -1)], dim=1)
        out.append(fr.to(torch.uint8) if bit_depth == 8 else fr.to(torch.int32))
    return out[0], out[1]


def synth_image_pair(H, W, seed):
    """Gray uint8 test / reference images [H, W] (numpy): seeded noise under a smooth pattern, the test a +-6 code perturbation of the
    reference.  Used where an input must be reproducible from its size and seed alone (goldens of large frames store only outputs)."""
    import numpy as np
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (H, W)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    ref = np.clip(0.6 * base + 50 + 40 * np.sin(xx / 7.0) * np.cos(yy / 5.0), 0, 255).astype(np.uint8)
    test = np.clip(ref.astype(np.int32) + rng.randint(-6, 7, (H, W)), 0, 255).astype(np.uint8)
    return test, ref
