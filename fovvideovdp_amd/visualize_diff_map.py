"""Colouring of difference maps (heat maps): the map modulates a tone-mapped, desaturated copy of the frame.
Torch version of the reference's pyfvvdp/visualize_diff_map.py (same results), kept as the public helper and as the
checker of the HIP colouring (`fvvdp_heatmap_colorize`) that `fvvdp.predict(..., heatmap=...)` uses."""
import torch

_LUMA = (0.212656, 0.715158, 0.072186)


def _interp1(x, v, x_q):
    """Piece-wise linear look-up with the reference's knot search and its +1e-6 in the denominator."""
    shp = x_q.shape
    q = x_q.flatten()
    imax = torch.bucketize(q, x).clamp_(max=x.shape[0] - 1)
    imin = (imax - 1).clamp_(0, x.shape[0] - 1)
    frc = (q - x[imin]) / (x[imax] - x[imin] + 0.000001)
    frc = torch.where(imax == imin, torch.zeros_like(frc), frc).clamp_(min=0.0)   # no masked writes: no host sync
    return (v[imin] * (1.0 - frc) + v[imax] * frc).reshape(shp)


def luminance_NCHW(x):
    if x.shape[1] == 3:
        return x[:, 0:1, ...] * _LUMA[0] + x[:, 1:2, ...] * _LUMA[1] + x[:, 2:3, ...] * _LUMA[2]
    return x


def log_luminance(x):
    y = luminance_NCHW(x)
    floor = torch.min(torch.where(y > 0.0, y, torch.full_like(y, float("inf"))))     # smallest positive value
    return torch.log(torch.maximum(y, floor))


def vis_tonemap(b, dr):
    """Histogram-based tone curve that fits log-luminance `b` into a dynamic range `dr` centred at 0.5."""
    t = 3.0
    b_min, b_max = torch.aminmax(b)
    if b.is_cuda:
        # device version without a host round trip: same histogram (1024 equal bins over [b_min, b_max], the last
        # one closed), built by scatter-add because torch.histc wants host scalars for its range; counts stay exact in fp32
        flat = (b - b_min) / (b_max - b_min + 1e-3) * dr + (1 - dr) / 2          # low-dynamic-range frames
        span = b_max - b_min
        pos = ((b.flatten() - b_min) * (1024.0 / torch.clamp(span, min=1e-30))).to(torch.int64).clamp_(0, 1023)
        b_p = torch.zeros(1024, dtype=b.dtype, device=b.device).scatter_add_(0, pos, torch.ones_like(pos, dtype=b.dtype))
        b_scale = b_min + span * torch.linspace(0.0, 1.0, 1024, device=b.device)
        b_p = b_p / torch.sum(b_p)
        dy = torch.pow(b_p, 1.0 / t)
        dy = dy / torch.sum(dy)
        v = torch.cumsum(dy, 0) * dr + (1.0 - dr) / 2.0
        return torch.where(span < dr, flat, _interp1(b_scale, v, b))
    if b_max - b_min < dr:
        return (b - b_min) / (b_max - b_min + 1e-3) * dr + (1 - dr) / 2
    b_scale = torch.linspace(b_min, b_max, 1024, device=b.device)
    b_p = torch.histc(b, 1024, b_min, b_max)
    b_p = b_p / torch.sum(b_p)
    dy = torch.pow(b_p, 1.0 / t)
    dy = dy / torch.sum(dy)
    v = torch.cumsum(dy, 0) * dr + (1.0 - dr) / 2.0
    return _interp1(b_scale, v, b)


_COLOR_MAPS = {
    "threshold": ([[0.2, 0.2, 1.0], [0.2, 1.0, 1.0], [0.2, 1.0, 0.2], [1.0, 1.0, 0.2], [1.0, 0.2, 0.2]],
                  [0.00, 0.25, 0.50, 0.75, 1.00]),
    "supra-threshold": ([[0.2, 1.0, 1.0], [1.0, 1.0, 1.0], [1.0, 1.0, 0.2]], [0.0, 0.5, 1.0]),
    "monochromatic": ([[1.0, 1.0, 1.0], [1.0, 1.0, 1.0]], [0.0, 1.0]),
}


_TABLE_CACHE = {}


def _color_tables(colormap_type, device):
    """Knots and luminance-normalised colours of a colour map on `device` (built once: a host->device upload per call
    would stall the stream for every frame)."""
    key = (colormap_type, str(device))
    if key not in _TABLE_CACHE:
        cm, cm_in = _COLOR_MAPS[colormap_type]
        color_map = torch.tensor(cm, device=device)
        color_map_in = torch.tensor(cm_in, device=device)
        color_map_l = color_map[:, 0:1] * _LUMA[0] + color_map[:, 1:2] * _LUMA[1] + color_map[:, 2:3] * _LUMA[2]
        _TABLE_CACHE[key] = (color_map_in, color_map / (torch.cat([color_map_l] * 3, 1) + 0.0001))
    return _TABLE_CACHE[key]


def color_tables_host(colormap_type):
    """(knots[K], colours[K,3]) of a colour map as fp32 numpy arrays for fvvdp_heatmap_colorize: the colours are divided
    by their luminance (+1e-4) with the same fp32 torch ops as visualize_diff_map."""
    color_map_in, color_map_ch = _color_tables(colormap_type, torch.device("cpu"))
    return color_map_in.numpy().astype("float32"), color_map_ch.numpy().astype("float32")


def visualize_diff_map(diff_map, context_image=None, type="pmap", colormap_type="supra-threshold"):
    """diff_map [N,1,H,W] in [0,1] -> sRGB-like [N,3,H,W]."""
    if colormap_type not in _COLOR_MAPS:
        raise RuntimeError("Unknown colormap: %s" % colormap_type)
    diff_map = torch.clamp(diff_map, 0.0, 1.0)
    if context_image is None:
        tmo_img = torch.ones_like(diff_map) * 0.5
    else:
        tmo_img = vis_tonemap(log_luminance(context_image), 0.6)
    color_map_in, color_map_ch = _color_tables(colormap_type, diff_map.device)
    cmap = torch.cat([_interp1(color_map_in, color_map_ch[:, k], diff_map) for k in range(3)], 1)
    return (cmap * torch.cat([tmo_img] * 3, dim=1)).clip(0., 1.)
