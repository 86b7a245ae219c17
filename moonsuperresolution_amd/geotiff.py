"""GeoTIFF raster I/O without GDAL — the data format either side of the hot path.

The reference reads band 1 of ``run-DRG.tif`` / ``run-DEM.tif`` with GDAL as float32 and keeps the geotransform and
projection (``loadImages``, process_full_tiles.py:158-182), and writes ``<map>_mean.tiff`` / ``_std.tiff`` (Float32)
and ``_good.tiff`` (UInt16) as LZW GeoTIFFs with ``PREDICTOR=2`` and a nodata tag (``saveGTiff``, :481-531).

``read_geotiff`` / ``write_geotiff`` do the same on the TIFF 6.0 / BigTIFF container directly:

* read: classic TIFF and BigTIFF, either byte order, strips or tiles, compression none / LZW / Deflate / PackBits,
  predictor 1 / 2 / 3, 8/16/32/64-bit (un)signed integer and IEEE float samples, chunky or planar multi-band files
  (band 1 is returned, like the reference); the result is float32.
* write: single band, little-endian, strips, LZW + horizontal predictor (or none / deflate), classic TIFF or BigTIFF
  chosen by size, GDAL_NODATA tag, and the GeoTIFF georeferencing tags of the input **passed through verbatim**
  (ModelPixelScale 33550, ModelTiepoint 33922, ModelTransformation 34264, GeoKeyDirectory 34735, GeoDoubleParams
  34736, GeoAsciiParams 34737, GDAL_METADATA 42112) — the output keeps the input's geotransform and projection
  byte for byte, which is what ``saveGTiff`` does with ``SetGeoTransform`` / ``SetProjection``.

The LZW codec lives in libmoonsr_hip.so's host code (csrc/host_codecs.cpp).  GDAL is absent here, so the codec and
container are validated against an independent implementation instead: PIL/libtiff reads what this module writes
and this module reads what PIL/libtiff writes (tests/test_geotiff.py).
"""
from __future__ import annotations

import ctypes as C
import struct
import zlib
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib

GEO_TAGS = (33550, 33922, 34264, 34735, 34736, 34737, 42112)
GDAL_NODATA = 42113
_TYPE_FMT = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 7: "B", 8: "h", 9: "i", 10: "ii", 11: "f", 12: "d",
             16: "Q", 17: "q", 18: "Q"}
_TYPE_SIZE = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8, 17: 8, 18: 8}


# ---- codecs ----------------------------------------------------------------------------------------------------------
def lzw_decode(data: bytes, out_size: int) -> np.ndarray:
    lib = _lib.load()
    src = np.frombuffer(data, np.uint8)
    out = np.empty(max(out_size, 1), np.uint8)
    n = lib.msr_lzw_decode(src.ctypes.data, src.size, out.ctypes.data, out_size)
    if n < 0:
        raise ValueError("malformed LZW stream")
    if n < out_size:
        out[n:out_size] = 0
    return out[:out_size]


def lzw_encode(data: np.ndarray) -> bytes:
    lib = _lib.load()
    src = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    out = np.empty(src.size + src.size // 2 + 1024, np.uint8)
    n = lib.msr_lzw_encode(src.ctypes.data, src.size, out.ctypes.data, out.size)
    if n < 0:
        raise ValueError("LZW encode failed")
    return out[:n].tobytes()


def _packbits_decode(data: bytes, out_size: int) -> np.ndarray:
    out = bytearray()
    i = 0
    while i < len(data) and len(out) < out_size:
        n = data[i] if data[i] < 128 else data[i] - 256
        i += 1
        if n >= 0:
            out += data[i:i + n + 1]
            i += n + 1
        elif n != -128:
            out += bytes([data[i]]) * (1 - n)
            i += 1
    return np.frombuffer(bytes(out[:out_size]).ljust(out_size, b"\0"), np.uint8)


# ---- reading ------------------------------------------------------------------------------------------------------------
def _read_ifd(buf: bytes, big: bool, bo: str, off: int) -> Dict[int, Tuple[int, int, bytes]]:
    """tag -> (type, count, raw value bytes) of the IFD at ``off``."""
    if big:
        n = struct.unpack_from(bo + "Q", buf, off)[0]
        pos, esz, vsz = off + 8, 20, 8
    else:
        n = struct.unpack_from(bo + "H", buf, off)[0]
        pos, esz, vsz = off + 2, 12, 4
    tags = {}
    for i in range(n):
        e = pos + i * esz
        tag, typ = struct.unpack_from(bo + "HH", buf, e)
        cnt = struct.unpack_from(bo + ("Q" if big else "I"), buf, e + 4)[0]
        size = _TYPE_SIZE.get(typ, 1) * cnt
        voff = e + 4 + vsz
        if size > vsz:
            voff = struct.unpack_from(bo + ("Q" if big else "I"), buf, voff)[0]
        tags[tag] = (typ, cnt, bytes(buf[voff:voff + size]))
    return tags


def _values(entry, bo: str):
    typ, cnt, raw = entry
    if typ == 2:
        return raw
    fmt = _TYPE_FMT[typ]
    vals = struct.unpack(bo + fmt * cnt if len(fmt) == 1 else bo + fmt * cnt, raw)
    return vals


def read_geotiff(path: str, band: int = 1):
    """Band ``band`` (1-based, like GDAL's GetRasterBand) of a (Geo)TIFF as float32, plus its metadata:
    ``{"geo": {tag: (type, count, raw bytes)}, "nodata": float | None, "dtype": numpy dtype of the file,
    "shape": (rows, cols), "byteorder": "<" | ">"}``.  Counterpart of loadImages (process_full_tiles.py:158-182)."""
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    head = bytes(buf[:16])
    if head[:2] == b"II":
        bo = "<"
    elif head[:2] == b"MM":
        bo = ">"
    else:
        raise ValueError(f"{path}: not a TIFF file")
    magic = struct.unpack_from(bo + "H", head, 2)[0]
    if magic == 42:
        big, ifd = False, struct.unpack_from(bo + "I", head, 4)[0]
    elif magic == 43:
        big, ifd = True, struct.unpack_from(bo + "Q", head, 8)[0]
    else:
        raise ValueError(f"{path}: not a TIFF file (magic {magic})")
    t = _read_ifd(buf, big, bo, ifd)

    def val(tag, default=None):
        return _values(t[tag], bo) if tag in t else default

    cols, rows = val(256)[0], val(257)[0]
    spp = val(277, (1,))[0]
    bps = val(258, (1,))[0]
    fmt = val(339, (1,))[0]
    comp = val(259, (1,))[0]
    pred = val(317, (1,))[0]
    planar = val(284, (1,))[0]
    if not 1 <= band <= spp:
        raise ValueError(f"band {band} out of range: the file has {spp}")
    kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
    if kind is None or bps not in (8, 16, 32, 64) or (kind == "f" and bps < 32):
        raise ValueError(f"{path}: unsupported sample format {fmt} / {bps} bits")
    dt = np.dtype(f"{bo}{kind}{bps // 8}")
    tiled = 322 in t
    if tiled:
        bw, bh = val(322)[0], val(323)[0]
        offs, cnts = val(324), val(325)
    else:
        bw, bh = cols, val(278, (rows,))[0]
        bh = min(bh, rows)
        offs, cnts = val(273), val(279)
    across, down = -(-cols // bw), -(-rows // bh)
    per_plane = across * down
    chunk_spp = 1 if planar == 2 else spp
    out = np.empty((rows, cols), np.float32)
    plane0 = (band - 1) * per_plane if planar == 2 else 0
    for by in range(down):
        for bx in range(across):
            i = plane0 + by * across + bx
            raw = bytes(buf[offs[i]:offs[i] + cnts[i]])
            h = bh if tiled else min(bh, rows - by * bh)
            nbytes = h * bw * chunk_spp * (bps // 8)
            if comp == 1:
                data = np.frombuffer(raw[:nbytes].ljust(nbytes, b"\0"), np.uint8)
            elif comp == 5:
                data = lzw_decode(raw, nbytes)
            elif comp in (8, 32946):
                data = np.frombuffer(zlib.decompress(raw)[:nbytes].ljust(nbytes, b"\0"), np.uint8)
            elif comp == 32773:
                data = _packbits_decode(raw, nbytes)
            else:
                raise ValueError(f"{path}: unsupported compression {comp}")
            if pred == 3:
                # floating-point predictor: bytes of a row are stored most-significant plane first, byte-differenced
                b = data.reshape(h, bw * chunk_spp * (bps // 8)).copy()
                b = np.cumsum(b, axis=1, dtype=np.uint8) if chunk_spp == 1 else _cumsum_stride(b, chunk_spp)
                nb = bps // 8
                b = b.reshape(h, nb, bw * chunk_spp).transpose(0, 2, 1)
                if bo == "<":
                    b = b[:, :, ::-1]
                arr = np.ascontiguousarray(b).view(dt).reshape(h, bw, chunk_spp)
            else:
                arr = data.view(dt).reshape(h, bw, chunk_spp)
                if pred == 2:
                    ut = np.dtype(f"u{bps // 8}")
                    arr = np.cumsum(arr.astype(dt.newbyteorder("=")).view(ut), axis=1, dtype=ut).view(dt.newbyteorder("="))
            ch = 0 if planar == 2 else band - 1
            y0, x0 = by * bh, bx * bw
            hh, ww = min(h, rows - y0), min(bw, cols - x0)
            out[y0:y0 + hh, x0:x0 + ww] = arr[:hh, :ww, ch]
    nodata = None
    if GDAL_NODATA in t:
        try:
            nodata = float(t[GDAL_NODATA][2].split(b"\0")[0].decode())
        except ValueError:
            nodata = None
    meta = {"geo": {k: t[k] for k in GEO_TAGS if k in t}, "nodata": nodata, "dtype": dt.newbyteorder("="),
            "shape": (rows, cols), "byteorder": bo}
    return out, meta


def _cumsum_stride(b: np.ndarray, stride: int) -> np.ndarray:
    out = b.copy()
    for s in range(stride, out.shape[1]):
        out[:, s] = (out[:, s].astype(np.uint16) + out[:, s - stride]).astype(np.uint8)
    return out


# ---- writing ------------------------------------------------------------------------------------------------------------
def write_geotiff(path: str, data: np.ndarray, meta: Optional[dict] = None, nodata: Optional[float] = None,
                  dtype=np.float32, compress: str = "lzw", predictor: int = 2, bigtiff: Optional[bool] = None) -> None:
    """Write a single-band GeoTIFF the way saveGTiff does (process_full_tiles.py:481-531): LZW, PREDICTOR=2,
    nodata tag, geotransform / projection taken from ``meta`` (as returned by ``read_geotiff``)."""
    if data.ndim != 2:
        raise ValueError("Data must be a 2-D array (the reference raises for rank != 2 too, process_full_tiles.py:505-519).")
    dt = np.dtype(dtype).newbyteorder("<")
    if dt.kind not in "uif" or dt.itemsize not in (1, 2, 4, 8):
        raise ValueError(f"unsupported dtype {dtype}")
    comp = {"none": 1, "lzw": 5, "deflate": 8}[compress]
    if comp == 1 or dt.itemsize == 8 and predictor == 2:
        predictor = 1
    rows, cols = data.shape
    arr = np.ascontiguousarray(data, dtype=dt)
    row_bytes = cols * dt.itemsize
    rps = max(1, min(rows, (1 << 16) // max(row_bytes, 1)))
    n_strips = -(-rows // rps)
    strips = []
    for s in range(n_strips):
        blk = arr[s * rps:(s + 1) * rps]
        if predictor == 2:
            u = blk.view(np.dtype(f"<u{dt.itemsize}"))
            d = u.copy()
            d[:, 1:] = u[:, 1:] - u[:, :-1]
            blk = d
        raw = np.ascontiguousarray(blk)
        strips.append(raw.tobytes() if comp == 1 else lzw_encode(raw) if comp == 5 else zlib.compress(raw.tobytes(), 6))
    total = sum(len(s) for s in strips)
    geo = dict((meta or {}).get("geo", {}))
    big = total + 8 * n_strips * 2 + sum(len(v[2]) for v in geo.values()) + 4096 > 0xFFFF0000
    if bigtiff is not None:
        big = bool(bigtiff)
    off_t = "Q" if big else "I"
    entries = {
        256: (4, 1, struct.pack("<I", cols)), 257: (4, 1, struct.pack("<I", rows)),
        258: (3, 1, struct.pack("<H", dt.itemsize * 8)), 259: (3, 1, struct.pack("<H", comp)),
        262: (3, 1, struct.pack("<H", 1)), 277: (3, 1, struct.pack("<H", 1)),
        278: (4, 1, struct.pack("<I", rps)), 284: (3, 1, struct.pack("<H", 1)),
        339: (3, 1, struct.pack("<H", {"u": 1, "i": 2, "f": 3}[dt.kind])),
    }
    if predictor != 1:
        entries[317] = (3, 1, struct.pack("<H", predictor))
    for k, v in geo.items():
        typ, cnt, raw = v
        if (meta or {}).get("byteorder", "<") == ">" and _TYPE_SIZE.get(typ, 1) > 1:
            w = _TYPE_SIZE[typ] if typ not in (5, 10) else 4
            raw = np.frombuffer(raw, np.dtype(f">u{w}")).astype(np.dtype(f"<u{w}")).tobytes()
        entries[k] = (typ, cnt, raw)
    if nodata is not None:
        txt = (repr(float(nodata)) if float(nodata) != int(nodata) else str(int(nodata))).encode() + b"\0"
        entries[GDAL_NODATA] = (2, len(txt), txt)
    header = 16 if big else 8
    pos = header
    offsets = []
    for s in strips:
        offsets.append(pos)
        pos += len(s) + (len(s) & 1)
    entries[273] = (16 if big else 4, n_strips, struct.pack("<" + off_t * n_strips, *offsets))
    entries[279] = (16 if big else 4, n_strips, struct.pack("<" + off_t * n_strips, *(len(s) for s in strips)))
    tags = sorted(entries)
    ifd_off = pos
    esz, vsz = (20, 8) if big else (12, 4)
    ifd_size = (8 if big else 2) + len(tags) * esz + (8 if big else 4)
    extra_off = ifd_off + ifd_size
    ifd = bytearray(struct.pack("<Q" if big else "<H", len(tags)))
    extra = bytearray()
    for k in tags:
        typ, cnt, raw = entries[k]
        ifd += struct.pack("<HH", k, typ) + struct.pack("<" + off_t, cnt)
        if len(raw) <= vsz:
            ifd += raw.ljust(vsz, b"\0")
        else:
            if len(extra) & 1:
                extra += b"\0"
            ifd += struct.pack("<" + off_t, extra_off + len(extra))
            extra += raw
    ifd += struct.pack("<" + off_t, 0)
    with open(path, "wb") as f:
        if big:
            f.write(b"II" + struct.pack("<HHHQ", 43, 8, 0, ifd_off))
        else:
            f.write(b"II" + struct.pack("<HI", 42, ifd_off))
        for s in strips:
            f.write(s)
            if len(s) & 1:
                f.write(b"\0")
        f.write(bytes(ifd))
        f.write(bytes(extra))


def geotransform(meta: dict) -> Optional[Tuple[float, float, float, float, float, float]]:
    """GDAL-style geotransform (x0, dx, 0, y0, 0, -dy) from ModelTiepoint + ModelPixelScale, if present."""
    geo = meta.get("geo", {})
    if 33550 not in geo or 33922 not in geo:
        return None
    bo = meta.get("byteorder", "<")
    sx, sy, _ = struct.unpack(bo + "3d", geo[33550][2][:24])
    i, j, _, x, y, _ = struct.unpack(bo + "6d", geo[33922][2][:48])
    return (x - i * sx, sx, 0.0, y + j * sy, 0.0, -sy)
