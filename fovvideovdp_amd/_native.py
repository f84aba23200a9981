"""ctypes binding of libfvvdp_hip.so (include/fvvdp_hip.h).  No fallback: if the library is missing or does not
load, every entry point raises -- the product never computes the hot path on the CPU."""
import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("FVVDP_LIB", os.path.join(_HERE, "libfvvdp_hip.so"))   # override: A/B builds
BUILD_FLAGS_NOTE = "hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -pragma-unroll-threshold=1000000"
SRC_PATH = os.path.join(_HERE, "csrc", "fvvdp_hip.hip")
INCLUDE_DIR = os.path.join(ROOT, "include")

FVVDP_U8, FVVDP_U16, FVVDP_F32 = 0, 1, 2
FVVDP_EUNSUPPORTED = -5
EOTF_LUT, EOTF_SRGB, EOTF_GAMMA, EOTF_PQ, EOTF_LINEAR, EOTF_ABSOLUTE, EOTF_NONE = range(7)
PSNR_SLICES = 256            # FVVDP_PSNR_SLICES
MAX_BANDS = 16
MAX_TAPS = 256
LUT_N = 32
RESIZE_MODES = {"nearest": 0, "bilinear": 1, "bicubic": 2, "area": 3}      # FVVDP_RESIZE_*


class Params(C.Structure):
    _fields_ = [("mask_p", C.c_float), ("mask_q", C.c_float * 2), ("mask_k", C.c_float), ("beta", C.c_float),
                ("sens_gain", C.c_float), ("lbkg_min", C.c_float), ("contrast_max", C.c_float), ("d_max", C.c_float)]


class Eotf(C.Structure):
    _fields_ = [("kind", C.c_int32), ("Y_peak", C.c_float), ("Y_black", C.c_float), ("gamma", C.c_float),
                ("L_min", C.c_float), ("L_max", C.c_float), ("d_lut", C.c_void_p)]


class YuvFormat(C.Structure):
    _fields_ = [("bit_depth", C.c_int32), ("chroma_420", C.c_int32), ("ycbcr2rgb", C.c_float * 9)]


class Geom(C.Structure):
    _fields_ = [("display_size_m", C.c_float * 2), ("distance_m", C.c_float), ("ppd_centre", C.c_float)]


class PoolParams(C.Structure):
    _fields_ = [("beta_sch", C.c_float), ("beta_tch", C.c_float), ("beta_t", C.c_float), ("w_transient", C.c_float),
                ("jod_a", C.c_float), ("beta_jod", C.c_float)]


class Pu21(C.Structure):
    _fields_ = [("p", C.c_float * 7), ("L_min", C.c_float), ("L_max", C.c_float)]


class BandMaps(C.Structure):
    _fields_ = [("d_D", C.c_void_p), ("d_contrast", C.c_void_p), ("d_lbkg", C.c_void_p), ("d_S", C.c_void_p)]


# every symbol declared in include/fvvdp_hip.h: name -> (restype, argtypes)
SYMBOLS = {
    "fvvdp_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(C.c_double), C.POINTER(Params)]),
    "fvvdp_ctx_destroy": (None, [C.c_void_p]),
    "fvvdp_last_error": (C.c_char_p, []),
    "fvvdp_ctx_level_size": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fvvdp_ctx_scratch_bytes": (C.c_size_t, [C.c_void_p]),
    "fvvdp_ctx_set_csf_1d": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "fvvdp_ctx_set_csf_3d": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "fvvdp_ctx_set_view_maps": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]),
    "fvvdp_temporal_channels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                          C.POINTER(Eotf), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fvvdp_temporal_channels_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int,
                                                 C.c_size_t, C.POINTER(Eotf), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                                 C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fvvdp_temporal_channels_yuv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(YuvFormat), C.c_size_t,
                                              C.POINTER(Eotf), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                              C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fvvdp_load_channels_planar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "fvvdp_bands_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float),
                                      C.POINTER(Geom), C.POINTER(BandMaps), C.c_void_p]),
    "fvvdp_bands_forward_pool": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float),
                                           C.POINTER(Geom), C.POINTER(BandMaps), C.POINTER(PoolParams), C.c_void_p, C.c_void_p]),
    "fvvdp_heatmap_reconstruct": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_float, C.c_float, C.c_float,
                                            C.c_void_p, C.c_void_p]),
    "fvvdp_export_level": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "fvvdp_pool_jod": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(PoolParams), C.c_void_p, C.c_void_p]),
    "fvvdp_heatmap_colorize": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                         C.POINTER(C.c_float), C.c_void_p, C.c_size_t, C.c_void_p]),
    "fvvdp_pu21_sse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t,
                                 C.POINTER(Eotf), C.POINTER(C.c_float), C.POINTER(Pu21), C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "fvvdp_yuv_frame_resized": (C.c_int, [C.c_void_p, C.POINTER(YuvFormat), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(Eotf), C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p]),
    "fvvdp_ctx_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "fvvdp_ctx_timing_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int, C.c_int]),
    "fvvdp_ctx_alloc_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "fvvdp_ctx_call_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
}

_lib = None


def build(force=False, verbose=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU).  Five translation units -- the band /
    pooling / side-metric kernels with the C ABI, and the temporal kernels once per sample type (temporal_launch.hip with
    -DK1_PART=0..3) -- are compiled concurrently and linked into one shared library."""
    csrc = os.path.dirname(SRC_PATH)
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc) if not f.startswith("_")] + [os.path.join(INCLUDE_DIR, "fvvdp_hip.h")]
    if not force and os.path.isfile(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(f) for f in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -pragma-unroll-threshold: the temporal kernels keep their filter window in registers and rely on FULL unrolling
    # of the tap loops (static ring slots); the default size cap silently falls back to scratch-memory indexing
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-pragma-unroll-threshold=1000000",
             "-I" + INCLUDE_DIR, "-I" + csrc] + os.environ.get("FVVDP_HIPCC_FLAGS", "").split()
    objdir = os.path.join(csrc, "_build")
    os.makedirs(objdir, exist_ok=True)
    units = [(SRC_PATH, [], os.path.join(objdir, "fvvdp_hip.o"))]
    units += [(os.path.join(csrc, "temporal_launch.hip"), ["-DK1_PART=%d" % k], os.path.join(objdir, "temporal_part%d.o" % k))
              for k in range(4)]
    procs = []
    for src, extra, obj in units:
        cmd = [hipcc] + flags + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(csrc, "exports.map")] + \
          [u[2] for u in units] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError("libfvvdp_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`. "
                               "There is no CPU fallback for the hot path." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)        # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(lib().fvvdp_last_error().decode("utf-8", "replace"))


def fptr(arr):
    """numpy fp32 array -> float*"""
    return arr.ctypes.data_as(C.POINTER(C.c_float))
