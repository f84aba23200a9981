"""Display models: photometry (display-encoded values -> cd/m^2) and geometry (pixels -> visual degrees).

Same class names, constructor arguments and methods as the reference (pyfvvdp/fvvdp_display_model.py) so user code
and user subclasses keep working.  These objects are configuration: on the hot path the stock classes are turned
into a small descriptor consumed by the HIP kernels (`native_eotf`, `native_geometry`); their torch `forward` is
used to build code-value look-up tables and by users who call it directly.
"""
import logging
import math
from abc import abstractmethod

import torch

from . import utils
from . import _native as nat


def srgb2lin(p):
    """sRGB non-linearity -> linear [0,1]."""
    return torch.where(p > 0.04045, ((p + 0.055) / 1.055) ** 2.4, p / 12.92)


def pq2lin(V):
    """SMPTE ST 2084 (PQ) code value [0,1] -> absolute cd/m^2 (up to 10000)."""
    Lmax, n, m = 10000, 0.15930175781250000, 78.843750000000000
    c1, c2, c3 = 0.83593750000000000, 18.851562500000000, 18.687500000000000
    im_t = torch.pow(V, 1 / m)
    return Lmax * torch.pow((im_t - c1).clamp(min=0) / (c2 - c3 * im_t), 1 / n)


class fvvdp_display_photometry:
    @abstractmethod
    def forward(self, V):
        pass

    @abstractmethod
    def print(self):
        pass

    @classmethod
    def list_displays(cls):
        models = utils.config_files.load("display_models.json")
        for display_name in models:
            cls.load(display_name).print()

    @classmethod
    def load(cls, display_name):
        models = utils.config_files.load("display_models.json")
        if display_name not in models:
            raise RuntimeError("Unknown display model: \"" + display_name + "\"")
        model = models[display_name]
        Y_peak = model["max_luminance"]
        if "min_luminance" in model:
            contrast = Y_peak / model["min_luminance"]
        else:
            contrast = model.get("contrast", 500)
        obj = fvvdp_display_photo_eotf(Y_peak, contrast=contrast, gamma=model.get("gamma", 2.2),
                                       EOTF=model.get("EOTF", "sRGB"), E_ambient=model.get("E_ambient", 0),
                                       k_refl=model.get("k_refl", 0.005), name=display_name)
        obj.full_name = model["name"]
        obj.short_name = display_name
        return obj


class _reflective_display(fvvdp_display_photometry):
    """Shared part of the EOTF and gain-gamma-offset models: black level raised by reflected ambient light."""

    def get_peak_luminance(self):
        return self.Y_peak

    def get_black_level(self):
        return self.E_ambient / math.pi * self.k_refl + self.Y_peak / self.contrast

    def print(self):
        Y_black = self.get_black_level()
        logging.info('Photometric display model: {}'.format(self.name))
        logging.info('  Peak luminance: {} cd/m^2'.format(self.Y_peak))
        if hasattr(self, "EOTF"):
            logging.info('  EOTF: {}'.format(self.EOTF))
        logging.info('  Contrast - theoretical: {}:1'.format(round(self.contrast)))
        logging.info('  Contrast - effective: {}:1'.format(round(self.Y_peak / Y_black)))
        logging.info('  Ambient light: {} lux'.format(self.E_ambient))
        logging.info('  Display reflectivity: {}%'.format(self.k_refl * 100))


class fvvdp_display_photo_eotf(_reflective_display):
    """SDR/HDR display with EOTF in {'sRGB','gamma','PQ','linear'}."""

    def __init__(self, Y_peak, contrast=1000, EOTF='sRGB', gamma=2.2, E_ambient=0, k_refl=0.005, name=None):
        self.Y_peak, self.contrast, self.EOTF, self.gamma = Y_peak, contrast, EOTF, gamma
        self.E_ambient, self.k_refl, self.name = E_ambient, k_refl, name

    def forward(self, V):
        if self.EOTF != 'linear' and (torch.any(V > 1).bool() or torch.any(V < 0).bool()):
            logging.warning("Pixel outside the valid range 0-1")
            V = V.clamp(0., 1.)
        Y_black = self.get_black_level()
        if self.EOTF == 'sRGB':
            return (self.Y_peak - Y_black) * srgb2lin(V) + Y_black
        if self.EOTF == 'gamma':
            return (self.Y_peak - Y_black) * torch.pow(V, self.gamma) + Y_black
        if self.EOTF == 'PQ':
            return pq2lin(V).clip(0.005, self.Y_peak) + Y_black
        if self.EOTF == 'linear':
            return V.clip(0.005, self.Y_peak) + Y_black
        raise RuntimeError(f"Unknown EOTF '{self.EOTF}'")


class fvvdp_display_photo_gog(_reflective_display):
    """Gain-gamma-offset SDR display (kept for compatibility); gamma == -1 selects the sRGB non-linearity."""

    def __init__(self, Y_peak, contrast=1000, gamma=2.2, E_ambient=0, k_refl=0.005, name=None):
        self.Y_peak, self.contrast, self.gamma = Y_peak, contrast, gamma
        self.E_ambient, self.k_refl, self.name = E_ambient, k_refl, name

    def forward(self, V):
        if torch.any(V > 1).bool() or torch.any(V < 0).bool():
            logging.warning("Pixel outside the valid range 0-1")
            V = V.clamp(0., 1.)
        Y_black = self.get_black_level()
        lin = srgb2lin(V) if self.gamma == -1 else torch.pow(V, self.gamma)
        return (self.Y_peak - Y_black) * lin + Y_black


class fvvdp_display_photo_absolute(fvvdp_display_photometry):
    """Content already in absolute cd/m^2; values are clamped to the display's range."""

    def __init__(self, L_max=10000, L_min=0.005):
        self.L_max, self.L_min = L_max, L_min

    def forward(self, V):
        L = V.clamp(self.L_min, self.L_max)
        if V.max() < 1:
            logging.warning('Pixel values are very low. Perhaps images are not scaled in the absolute units of cd/m^2.')
        return L

    def get_peak_luminance(self):
        return self.L_max

    def get_black_level(self):
        return self.L_min

    def print(self):
        logging.info('Photometric display model:')
        logging.info('  Absolute photometric/colorimetric values')


def native_eotf(photometry):
    """(kind, dict) descriptor for the HIP kernels when `photometry` is exactly one of the stock classes with a
    closed-form model on float input, else None (the caller then tabulates `forward` or calls it per frame)."""
    t = type(photometry)
    if t is fvvdp_display_photo_eotf:
        kinds = {"sRGB": nat.EOTF_SRGB, "gamma": nat.EOTF_GAMMA, "PQ": nat.EOTF_PQ, "linear": nat.EOTF_LINEAR}
        if photometry.EOTF not in kinds:
            raise RuntimeError(f"Unknown EOTF '{photometry.EOTF}'")
        return kinds[photometry.EOTF], dict(Y_peak=photometry.Y_peak, Y_black=photometry.get_black_level(),
                                            gamma=photometry.gamma)
    if t is fvvdp_display_photo_gog:
        kind = nat.EOTF_SRGB if photometry.gamma == -1 else nat.EOTF_GAMMA
        return kind, dict(Y_peak=photometry.Y_peak, Y_black=photometry.get_black_level(), gamma=photometry.gamma)
    if t is fvvdp_display_photo_absolute:
        return nat.EOTF_ABSOLUTE, dict(L_min=photometry.L_min, L_max=photometry.L_max)
    return None


class fvvdp_display_geometry:
    """Resolution, physical size and viewing distance -> pixels per degree, view directions, magnification."""

    def __init__(self, resolution, distance_m=None, distance_display_heights=None, fov_horizontal=None,
                 fov_vertical=None, fov_diagonal=None, diagonal_size_inches=None) -> None:
        self.resolution = resolution
        self.fixed_ppd = None
        ar = resolution[0] / resolution[1]
        if diagonal_size_inches is not None:
            height_mm = math.sqrt((diagonal_size_inches * 25.4) ** 2 / (1 + ar ** 2))
            self.display_size_m = (ar * height_mm / 1000, height_mm / 1000)
        if distance_m is not None and distance_display_heights is not None:
            raise RuntimeError("You can pass only one of: 'distance_m', 'distance_display_heights'.")
        fov_given = [f is not None for f in (fov_horizontal, fov_vertical, fov_diagonal)]
        if distance_m is not None:
            self.distance_m = distance_m
        elif distance_display_heights is not None:
            if not hasattr(self, "display_size_m"):
                raise RuntimeError("You need to specify display diagonal size 'diagonal_size_inches' to specify "
                                   "viewing distance as 'distance_display_heights' ")
            self.distance_m = distance_display_heights * self.display_size_m[1]
        elif any(fov_given):
            self.distance_m = 3          # default viewing distance for VR headsets
        else:
            raise RuntimeError("Viewing distance must be specified as 'distance_m' or 'distance_display_heights'.")
        if sum(fov_given) > 1:
            raise RuntimeError("You can pass only one of 'fov_horizontal', 'fov_vertical', 'fov_diagonal'. The other "
                               "dimensions are inferred from the resolution assuming that the pixels are square.")
        if fov_horizontal is not None:
            width_m = 2 * math.tan(math.radians(fov_horizontal / 2)) * self.distance_m
            self.display_size_m = (width_m, width_m / ar)
        elif fov_vertical is not None:
            height_m = 2 * math.tan(math.radians(fov_vertical / 2)) * self.distance_m
            self.display_size_m = (height_m * ar, height_m)
        elif fov_diagonal is not None:
            # work on a distance measure: degrees do not obey Pythagoras
            distance_px = math.hypot(resolution[0], resolution[1]) / (2.0 * math.tan(math.radians(fov_diagonal * 0.5)))
            height_deg = math.degrees(math.atan(resolution[1] / 2 / distance_px)) * 2
            height_m = 2 * math.tan(math.radians(height_deg / 2)) * self.distance_m
            self.display_size_m = (height_m * ar, height_m)
        self.display_size_deg = tuple(2 * math.degrees(math.atan(s / (2 * self.distance_m))) for s in self.display_size_m)
        self.ppd_centre = 1 / (2 * math.degrees(math.atan(0.5 * self.display_size_m[0] / resolution[0] / self.distance_m)))

    def get_ppd(self, view_dir=None):
        """Pixels per degree at the screen centre, or for view directions [2,h,w] (degrees)."""
        if view_dir is None:
            return self.ppd_centre
        view_angle = torch.sqrt(torch.sum(view_dir ** 2, dim=0, keepdim=False))
        view_angle = torch.minimum(view_angle, torch.tensor(89.9))
        delta = (1 / self.ppd_centre) / 2
        tan_delta = math.tan(math.radians(delta))
        tan_a = torch.tan(torch.deg2rad(view_angle))
        return self.ppd_centre * (torch.tan(torch.deg2rad(view_angle + delta)) - tan_a) / tan_delta

    def pix2view_direction(self, resolution_pix, x_pix, y_pix):
        """Pixel positions (top-left origin) -> view direction in degrees, x right, y up, [2,h,w]."""
        shift_to_centre = -resolution_pix / 2
        x_m = (x_pix + shift_to_centre[0]) * self.display_size_m[0] / resolution_pix[0]
        y_m = -(y_pix + shift_to_centre[1]) * self.display_size_m[1] / resolution_pix[1]
        return torch.stack((torch.rad2deg(torch.atan(x_m / self.distance_m)),
                            torch.rad2deg(torch.atan(y_m / self.distance_m))), dim=0)

    def get_resolution_magnification(self, view_dir):
        """ppd(view_dir)/ppd(centre): how the angular resolution grows away from the screen centre."""
        if self.fixed_ppd is not None:
            return torch.ones((), device=view_dir.device)
        return self.get_ppd(view_dir) / self.get_ppd()

    def print(self):
        logging.info('Geometric display model:')
        logging.info('  Resolution: {w} x {h} pixels'.format(w=self.resolution[0], h=self.resolution[1]))
        logging.info('  Display size: {w:.1f} x {h:.1f} cm'.format(w=self.display_size_m[0] * 100, h=self.display_size_m[1] * 100))
        logging.info('  Display size: {w:.2f} x {h:.2f} deg'.format(w=self.display_size_deg[0], h=self.display_size_deg[1]))
        logging.info('  Viewing distance: {d:.3f} m'.format(d=self.distance_m))
        logging.info('  Pixels-per-degree (center): {ppd:.2f}'.format(ppd=self.get_ppd()))

    @classmethod
    def load(cls, display_name):
        models = utils.config_files.load("display_models.json")
        if display_name not in models:
            raise RuntimeError("Error: Display model '%s' not found in display_models.json" % display_name)
        model = models[display_name]
        assert "resolution" in model
        W, H = model["resolution"]
        if "viewing_distance_meters" in model:
            distance_m = model["viewing_distance_meters"]
        elif "viewing_distance_inches" in model:
            distance_m = model["viewing_distance_inches"] * 0.0254
        else:
            distance_m = None
        if "diagonal_size_meters" in model:
            diag_inch = model["diagonal_size_meters"] / 0.0254
        else:
            diag_inch = model.get("diagonal_size_inches")
        return cls((W, H), distance_m=distance_m, fov_diagonal=model.get("fov_diagonal"), diagonal_size_inches=diag_inch)


def native_geometry(geometry):
    """Descriptor for the in-kernel foveated maps when `geometry` is exactly the stock class, else None."""
    if type(geometry) is fvvdp_display_geometry and geometry.fixed_ppd is None:
        return dict(display_size_m=geometry.display_size_m, distance_m=geometry.distance_m, ppd_centre=geometry.ppd_centre)
    return None


def photometry_state(photometry):
    """Value key of a STOCK photometry object (every attribute its `forward` reads), or None for user classes whose
    state cannot be enumerated."""
    t = type(photometry)
    if t is fvvdp_display_photo_eotf:
        return ("eotf", float(photometry.Y_peak), float(photometry.contrast), str(photometry.EOTF), float(photometry.gamma),
                float(photometry.E_ambient), float(photometry.k_refl))
    if t is fvvdp_display_photo_gog:
        return ("gog", float(photometry.Y_peak), float(photometry.contrast), float(photometry.gamma),
                float(photometry.E_ambient), float(photometry.k_refl))
    if t is fvvdp_display_photo_absolute:
        return ("abs", float(photometry.L_min), float(photometry.L_max))
    return None


class code_value_tables:
    """Device-resident luminance of every integer code value (256 or 65536 entries) through a display model's own
    `forward` (so user photometry subclasses work).

    Entries are keyed by VALUE, never by object identity (CPython reuses `id()` after garbage collection): stock
    classes by the attributes their `forward` reads; any other class is re-tabulated on the host on every request
    and the cached device copy is reused only when the fresh table is bit-identical to it."""

    MAX_ENTRIES = 16

    def __init__(self):
        self._stock = {}        # (state tuple, nbits) -> device tensor
        self._user = {}         # nbits -> (host table, device tensor)
        self._range = {}        # data_ptr of a device table -> (smallest, largest entry), from the host copy

    def clear(self):
        self._stock.clear()
        self._user.clear()
        self._range.clear()

    def table_range(self, table):
        """(smallest, largest) luminance in a table returned by get() -- (0, 0) if unknown or not finite."""
        return self._range.get(table.data_ptr(), (0.0, 0.0))

    def _note_range(self, host, dev):
        lo, hi = float(host.min()), float(host.max())
        if len(self._range) > 4 * self.MAX_ENTRIES:
            self._range.clear()
        self._range[dev.data_ptr()] = (lo, hi) if (math.isfinite(lo) and math.isfinite(hi) and hi > lo >= 0.0) else (0.0, 0.0)
        return dev

    @staticmethod
    def _tabulate(photometry, nbits):
        n = 1 << nbits
        codes = torch.arange(n, dtype=torch.int32).to(torch.float32) / float(n - 1)
        return photometry.forward(codes.view(1, 1, 1, 1, n)).reshape(-1).to(torch.float32).contiguous().cpu()

    def get(self, photometry, nbits, device):
        state = photometry_state(photometry)
        if state is not None:
            key = (state, nbits, str(device))
            if key not in self._stock:
                if len(self._stock) >= self.MAX_ENTRIES:
                    self._stock.clear()
                host = self._tabulate(photometry, nbits)
                self._stock[key] = self._note_range(host, host.to(device))
            return self._stock[key]
        host = self._tabulate(photometry, nbits)
        key = (nbits, str(device))
        hit = self._user.get(key)
        if hit is None or not torch.equal(hit[0], host):
            hit = (host, self._note_range(host, host.to(device)))
            self._user[key] = hit
        return hit[1]
