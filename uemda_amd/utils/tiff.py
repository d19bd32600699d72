"""Minimal baseline-TIFF reader / writer for the superpixel id maps (`<name>.tif`, one int32 per pixel, written by
`skimage.io.imsave` = tifffile without compression in the reference, gast/superpixels.py:113-114,148-149).

Handles what that writer produces and a little more: little- or big-endian classic TIFF, one sample per pixel,
8/16/32-bit unsigned / signed integer or 32-bit float samples, any strip layout, no compression.  Anything else
raises (no silent misread)."""
import struct

import numpy as np

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 8: ("h", 2), 9: ("i", 4), 16: ("Q", 8)}


def _values(buf, bo, typ, count, field_bytes, field_off):
    fmt, size = _TYPES[typ]
    total = size * count
    data = field_bytes if total <= 4 else buf[field_off:field_off + total]
    if typ == 5:
        v = struct.unpack(bo + "II" * count, data[:total])
        return [v[2 * i] / max(v[2 * i + 1], 1) for i in range(count)]
    return list(struct.unpack(bo + fmt * count, data[:total]))


def read_tiff(path):
    buf = open(path, "rb").read()
    if buf[:2] == b"II":
        bo = "<"
    elif buf[:2] == b"MM":
        bo = ">"
    else:
        raise ValueError(f"{path}: not a TIFF file")
    magic, ifd = struct.unpack(bo + "HI", buf[2:8])
    if magic != 42:
        raise ValueError(f"{path}: only classic TIFF (magic 42) is supported, got {magic}")
    n = struct.unpack(bo + "H", buf[ifd:ifd + 2])[0]
    tags = {}
    for i in range(n):
        e = buf[ifd + 2 + 12 * i: ifd + 14 + 12 * i]
        tag, typ, count = struct.unpack(bo + "HHI", e[:8])
        if typ not in _TYPES:
            continue
        off = struct.unpack(bo + "I", e[8:12])[0]
        tags[tag] = _values(buf, bo, typ, count, e[8:12], off)
    width, height = tags[256][0], tags[257][0]
    bits = tags.get(258, [1])
    spp = tags.get(277, [1])[0]
    compression = tags.get(259, [1])[0]
    sample_format = tags.get(339, [1])[0]
    if spp != 1 or len(bits) != 1:
        raise ValueError(f"{path}: only single-sample images are supported (SamplesPerPixel={spp})")
    if compression != 1:
        raise ValueError(f"{path}: compressed TIFF (Compression={compression}) is not supported")
    kind = {1: "u", 2: "i", 3: "f"}.get(sample_format)
    if kind is None or bits[0] not in (8, 16, 32) or (kind == "f" and bits[0] != 32):
        raise ValueError(f"{path}: unsupported sample format {sample_format} with {bits[0]} bits")
    dtype = np.dtype(f"{bo}{kind}{bits[0] // 8}")
    offsets, counts = tags[273], tags.get(279)
    rows_per_strip = tags.get(278, [height])[0]
    row_bytes = width * dtype.itemsize
    out = np.empty((height, width), dtype=dtype.newbyteorder("="))
    row = 0
    for si, off in enumerate(offsets):
        rows = min(rows_per_strip, height - row)
        nbytes = rows * row_bytes
        if counts is not None and counts[si] < nbytes:
            raise ValueError(f"{path}: strip {si} is shorter than its rows")
        out[row:row + rows] = np.frombuffer(buf, dtype=dtype, count=rows * width, offset=off).reshape(rows, width)
        row += rows
    if row != height:
        raise ValueError(f"{path}: strips cover {row} of {height} rows")
    return out


def write_tiff(path, arr):
    """(H,W) integer / float32 array -> little-endian, uncompressed, single-strip TIFF."""
    arr = np.ascontiguousarray(arr)
    if arr.ndim != 2:
        raise ValueError("write_tiff: expected a 2-D array")
    kind = {"u": 1, "i": 2, "f": 3}.get(arr.dtype.kind)
    if kind is None or arr.dtype.itemsize not in (1, 2, 4) or (kind == 3 and arr.dtype.itemsize != 4):
        raise ValueError(f"write_tiff: unsupported dtype {arr.dtype}")
    data = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
    h, w = arr.shape
    entries = [(256, 4, w), (257, 4, h), (258, 3, arr.dtype.itemsize * 8), (259, 3, 1), (262, 3, 1), (273, 4, 8),
               (277, 3, 1), (278, 4, h), (279, 4, len(data)), (339, 3, kind)]
    ifd_off = 8 + len(data) + (len(data) & 1)
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, ifd_off))
        f.write(data)
        if len(data) & 1:
            f.write(b"\x00")
        f.write(struct.pack("<H", len(entries)))
        for tag, typ, val in entries:
            f.write(struct.pack("<HHI", tag, typ, 1))
            f.write(struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
        f.write(struct.pack("<I", 0))
