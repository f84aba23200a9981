"""File readers around the metric -- NOT part of the accelerated path (out of scope): a PNG reader
for `load_image_as_array` and a source class for raw planar `.yuv` files whose names encode their format.  They live with the
examples; the product takes arrays (`fvvdp.predict`), raw planar YUV frames (`fvvdp_video_source_yuv_frames`) or a user's
own `fvvdp_video_source`."""
import logging
import os
import struct
import zlib

import numpy as np


def _unfilter(raw, H, stride, bpp):
    """PNG scanline filters 0-4 (RFC 2083 section 6).  Sub / Up are vectorised; Average / Paeth run the serial recurrence
    per byte column group."""
    out = np.zeros((H, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    p = 0
    for y in range(H):
        ft = raw[p]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=p + 1).astype(np.int32)
        p += 1 + stride
        if ft == 0:
            cur = line
        elif ft == 1:
            cur = (np.cumsum(line.reshape(-1, bpp), axis=0) & 0xFF).reshape(-1)
        elif ft == 2:
            cur = (line + prev) & 0xFF
        elif ft in (3, 4):
            cur = np.zeros(stride, dtype=np.int32)
            ln, pv = line.reshape(-1, bpp), prev.reshape(-1, bpp)
            cr = cur.reshape(-1, bpp)
            a = np.zeros(bpp, dtype=np.int32)
            c = np.zeros(bpp, dtype=np.int32)
            for i in range(ln.shape[0]):              # serial along the row, vectorised over the bytes of a pixel
                b = pv[i]
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    pa, pb, pc = np.abs(b - c), np.abs(a - c), np.abs(a + b - 2 * c)
                    pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
                a = (ln[i] + pred) & 0xFF
                cr[i] = a
                c = b
        else:
            raise RuntimeError("corrupt PNG: unknown filter type %d" % ft)
        out[y] = cur
        prev = cur
    return out


def read_png(path):
    """8 / 16-bit non-interlaced PNG -> uint8 / uint16 array [H, W, C]."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise RuntimeError("%s is not a PNG file" % path)
    pos, idat, ihdr = 8, [], None
    while pos + 8 <= len(data):
        (ln,) = struct.unpack(">I", data[pos:pos + 4])
        typ = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + ln]
        pos += 12 + ln
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    if ihdr is None:
        raise RuntimeError("corrupt PNG: no IHDR")
    W, H, depth, ctype, _, _, interlace = ihdr
    if interlace != 0 or depth not in (8, 16) or ctype not in (0, 2, 4, 6):
        raise RuntimeError("PNG variant not supported by the built-in reader (interlaced, palette or < 8 bit); install imageio")
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    bpp = ch * depth // 8
    out = _unfilter(zlib.decompress(b"".join(idat)), H, W * bpp, bpp)
    if depth == 16:
        img = out.reshape(H, W, ch, 2).astype(np.uint16)
        return (img[..., 0] << 8) | img[..., 1]
    return out.reshape(H, W, ch)


def load_image_as_array(imgfile):
    """Image file -> numpy array [H, W, C] (uint8 / uint16 / float32), extra channels beyond RGB dropped, gray images
    expanded to [H, W, 1]: the contract of the reference's function."""
    ext = os.path.splitext(imgfile)[1].lower()
    if ext == ".png":
        try:
            img = read_png(imgfile)
            if img.shape[2] in (2, 4):            # alpha
                logging.warning(f'Input image {imgfile} has an alpha channel. Ignoring it.')
                img = img[:, :, :img.shape[2] - 1]
            return img
        except RuntimeError:
            pass                                  # fall through to imageio for exotic PNG variants
    try:
        import imageio.v2 as io
    except ImportError as e:
        raise RuntimeError("Reading %s needs the 'imageio' package (only PNG is built in)" % imgfile) from e
    img = np.asarray(io.imread(imgfile))
    if img.ndim == 3 and img.shape[2] > 3:
        logging.warning(f'Input image {imgfile} has more than 3 channels (alpha?). Ignoring the extra channels.')
        img = img[:, :, :3]
    if img.ndim == 2:
        img = img[:, :, np.newaxis]
    return img


import torch

from fovvideovdp_amd.video_source_yuv import fvvdp_video_source_yuv_frames


# ---- raw .yuv files (no container, no decoder) -----------------------------------------------------------------
def decode_video_props(fname):
    """Video properties from a raw-YUV file name of the form `<name>_<W>x<H>_<8b|10b>_<420|444>_<709|2020>_<fps>fps.yuv`
    (same convention and defaults as the reference's pyfvvdp/video_source_yuv.py:6-53: underscore-separated fields in
    any order, missing ones default to 1920x1080, 24 fps, 8 bit, '2020', '420').  A trailing 'p' on the height
    (`1280x720p`), which the reference's parser trips over, is accepted."""
    import os
    import re
    props = dict(width=1920, height=1080, fps=24, bit_depth=8, color_space='2020', chroma_ss='420')
    fields = os.path.splitext(os.path.basename(fname))[0].split("_")
    for f in fields:
        m = re.fullmatch(r'(\d+)x(\d+)p?', f)
        if m:
            props["width"], props["height"] = int(m.group(1)), int(m.group(2))
        elif f.endswith("fps"):
            props["fps"] = float(f[:-3])
        elif f in ("444", "420"):
            props["chroma_ss"] = f
        elif f in ("10", "10b"):
            props["bit_depth"] = 10
        elif f in ("8", "8b"):
            props["bit_depth"] = 8
        elif f in ("2020", "ct2020", "pq2020"):
            props["color_space"] = "2020"
        elif f in ("709", "bt709"):
            props["color_space"] = "709"
    return props


def create_yuv_fname(basename, vprops):
    """File name that encodes the properties (pyfvvdp/video_source_yuv.py:55-64)."""
    fps = vprops["fps"]
    fps = round(fps, 3) if round(fps) != fps else int(fps)
    return "%s_%dx%d_%db_%s_%s_%sfps.yuv" % (basename, vprops["width"], vprops["height"], vprops["bit_depth"],
                                             vprops["chroma_ss"], vprops["color_space"], fps)


class fvvdp_video_source_yuv_file(fvvdp_video_source_yuv_frames):
    """Test / reference pair of raw planar .yuv files (the reference's `fvvdp_video_source_yuv_file`,
    pyfvvdp/video_source_yuv.py:238-292, whose constructor cannot run: it logs attributes its reader does not have).
    The frames are read into host memory as they are stored (1.5 or 3 bytes per pixel and stream for 8 bit), uploaded in
    that form and unpacked by the fused ingest kernel.  With `full_screen_resize` (the CLI's --full-screen-resize) every frame is resized in RGB
    to `resize_resolution` on the GPU (`fvvdp_video_source_yuv_frames`, `fvvdp_yuv_frame_resized`) and the metric takes luminance frames from there."""

    def __init__(self, test_fname, reference_fname, display_photometry='standard_4k', color_space_name='auto', frames=-1,
                 full_screen_resize=None, resize_resolution=None, verbose=False):
        import os
        tp, rp = decode_video_props(test_fname), decode_video_props(reference_fname)
        for k in ("width", "height", "bit_depth", "chroma_ss", "color_space", "fps"):   # one YCbCr matrix / frame rate for both
            if tp[k] != rp[k]:
                raise RuntimeError("Test and reference .yuv files differ in %s (%s vs %s)" % (k, tp[k], rp[k]))
        for fn in (test_fname, reference_fname):
            if not os.path.isfile(fn):
                raise FileNotFoundError("File {} not found".format(fn))
        dtype = np.uint16 if tp["bit_depth"] > 8 else np.uint8
        uv = (tp["width"] // 2) * (tp["height"] // 2) if tp["chroma_ss"] == "420" else tp["width"] * tp["height"]
        elems = tp["width"] * tp["height"] + 2 * uv

        def load(fn):
            mm = np.memmap(fn, dtype, mode="r")
            n = mm.shape[0] // elems                       # whole frames only, like the reference's frame_count
            n = n if frames == -1 else min(n, frames)
            return np.array(mm[:n * elems]).reshape(n, elems)

        t, r = load(test_fname), load(reference_fname)
        n = min(t.shape[0], r.shape[0])
        if n < 1:
            raise RuntimeError("The .yuv files hold no complete frame")
        super().__init__(t[:n], r[:n], tp["fps"], tp["width"], tp["height"], bit_depth=tp["bit_depth"],
                         chroma_ss=tp["chroma_ss"], color_space="bt2020nc" if tp["color_space"] == "2020" else "bt709",
                         display_photometry=display_photometry, color_space_name=color_space_name,
                         full_screen_resize=full_screen_resize, resize_resolution=resize_resolution)
