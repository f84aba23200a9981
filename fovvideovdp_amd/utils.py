"""Configuration lookup and small helpers (host side, not on the hot path).

Mirrors the behaviour of the reference's `utils.config_files` (pyfvvdp/utils.py:129-154): a file is looked up in
the directory given to `set_config_dir`, then in `$FVVDP_PATH`, then in the packaged defaults.  The packaged
defaults live in one consolidated JSON (`data/defaults.json`, sections keyed by the reference's file names) and
the CSF look-up tables in `data/csf_lut.npz` (see tools/import_reference_data.py).
"""
import json
import os

import numpy as np

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_defaults_cache = None


def json2dict(file):
    if not os.path.isfile(file):
        raise RuntimeError(f"Error: Cannot find file {file}")
    with open(file, "r") as f:
        return json.load(f)


def _defaults():
    global _defaults_cache
    if _defaults_cache is None:
        _defaults_cache = json2dict(os.path.join(_DATA_DIR, "defaults.json"))
    return _defaults_cache


class config_files:
    fvvdp_config_dir = None

    @classmethod
    def set_config_dir(cls, path):
        cls.fvvdp_config_dir = path

    @classmethod
    def _user_file(cls, fname):
        for d in (cls.fvvdp_config_dir, os.getenv("FVVDP_PATH")):
            if d is not None:
                path = os.path.join(d, fname)
                if os.path.isfile(path):
                    return path
        return None

    @classmethod
    def find(cls, fname):
        """Path of a user-supplied configuration file, or of the packaged defaults when there is none."""
        path = cls._user_file(fname)
        if path is not None:
            return path
        if fname in _defaults():
            return os.path.join(_DATA_DIR, "defaults.json")
        raise RuntimeError(f"The configuration file {fname} not found")

    @classmethod
    def load(cls, fname):
        """Parsed content of configuration file `fname` (user directory, $FVVDP_PATH, packaged defaults)."""
        path = cls._user_file(fname)
        if path is not None:
            return json2dict(path)
        d = _defaults()
        if fname in d:
            return d[fname]
        raise RuntimeError(f"The configuration file {fname} not found")


def csf_cache_key(omega, sigma, k_cm):
    """Name of the pre-computed CSF table for temporal frequency omega (same key as pyfvvdp/fvvdp.py:502-503)."""
    return ("o%g_s%g_cm%f" % (omega, sigma, k_cm)).replace("-", "n").replace(".", "_")


def load_csf_lut(omega, sigma, k_cm, cache_dirs=()):
    """dict of fp32 numpy arrays: S_log[32(Y),32(rho),32(ecc)], Y_log, rho_log, ecc_sqrt, Y, rho, ecc."""
    key = csf_cache_key(omega, sigma, k_cm)
    for d in cache_dirs:
        fname = os.path.join(d, key + "_gpu0.mat")
        if os.path.isfile(fname):                          # user-supplied MATLAB cache file
            import scipy.io as spio
            m = spio.loadmat(fname, struct_as_record=False, squeeze_me=True)["lut"]
            return {f: np.ascontiguousarray(getattr(m, f), dtype=np.float32) for f in m._fieldnames}
    z = np.load(os.path.join(_DATA_DIR, "csf_lut.npz"))
    names = [n for n in z.files if n.startswith(key + "/")]
    if not names:
        raise RuntimeError("Error: cache file for %s not found" % key)
    return {n.split("/", 1)[1]: np.ascontiguousarray(z[n], dtype=np.float32) for n in names}
