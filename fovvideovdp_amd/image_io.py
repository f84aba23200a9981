"""Image loading for the examples-style entry point `load_image_as_array` (reference: pyfvvdp/video_source_file.py:29-55,
which needs imageio + the FreeImage plugin for 16-bit PNG).  Here PNG files (8 / 16 bit, gray / gray+alpha / RGB / RGBA,
non-interlaced) are decoded with the standard library only; other formats go to imageio when it is installed."""
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
