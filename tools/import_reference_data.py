#!/usr/bin/env python3
"""Convert the reference's *data* files (not code) into the package's own data format.

Runs only in the build container (needs /root/reference).  Reads
  pyfvvdp/fvvdp_data/{fvvdp_parameters,fvvdp_parameters_1_0,display_models,color_spaces}.json
  pyfvvdp/csf_cache/o{0,5}_sn1_5_cm0_604562_gpu0.mat        (pre-computed CSF LUTs, MAT v5)
and writes
  fovvideovdp_amd/data/defaults.json   one consolidated JSON, sections keyed by the reference file name
  fovvideovdp_amd/data/csf_lut.npz     arrays "<key>/<field>" with key = reference cache key (fvvdp.py:502-503)

The calibration constants and the LUT are inputs of the metric: there is no
way to re-derive them (the LUT generator is MATLAB code, matlab/utils/CSF_st_fov.m:107-130).
"""
import json
import os
import sys

import numpy as np
import scipy.io as spio

REF = os.environ.get("FVVDP_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fovvideovdp_amd", "data")


def main():
    os.makedirs(OUT, exist_ok=True)
    sections = {}
    for name in ("fvvdp_parameters.json", "fvvdp_parameters_1_0.json", "display_models.json", "color_spaces.json"):
        with open(os.path.join(REF, "pyfvvdp", "fvvdp_data", name), "r") as f:
            sections[name] = json.load(f)
    with open(os.path.join(OUT, "defaults.json"), "w", encoding="utf-8") as f:
        json.dump(sections, f, indent=1, sort_keys=True)

    arrays = {}
    cache_dir = os.path.join(REF, "pyfvvdp", "csf_cache")
    for fn in sorted(os.listdir(cache_dir)):
        if not fn.endswith("_gpu0.mat"):
            continue
        key = fn[: -len("_gpu0.mat")]
        m = spio.loadmat(os.path.join(cache_dir, fn), struct_as_record=False, squeeze_me=True)
        lut = m["lut"]
        for field in lut._fieldnames:
            arrays[key + "/" + field] = np.ascontiguousarray(getattr(lut, field), dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "csf_lut.npz"), **arrays)
    print("wrote", os.path.join(OUT, "defaults.json"), "and csf_lut.npz with", sorted(arrays))


if __name__ == "__main__":
    sys.exit(main())
