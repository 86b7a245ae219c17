"""CPU: GeoTIFF reader / writer cross-validated against PIL/libtiff, an independent codec
(GDAL, which the reference uses for loadImages / saveGTiff, is not installed here)."""
import struct

import numpy as np
import pytest
from PIL import Image, features

from moonsuperresolution_amd import geotiff as G
from tests.helpers import synthetic_raster

pytestmark = pytest.mark.skipif(not features.check("libtiff"), reason="PIL without libtiff")


def _geo_meta():
    scale = struct.pack("<3d", 5.0, 5.0, 0.0)
    tie = struct.pack("<6d", 0, 0, 0, 1000.0, -2000.0, 0)
    keys = struct.pack("<8H", 1, 1, 0, 1, 1024, 0, 1, 1)
    return {"geo": {33550: (12, 3, scale), 33922: (12, 6, tie), 34735: (3, 8, keys),
                    34737: (2, 12, b"MOON_2000|\x00\x00")}, "byteorder": "<"}


@pytest.mark.parametrize("compress,predictor", [("lzw", 2), ("lzw", 1), ("none", 1), ("deflate", 2)])
@pytest.mark.parametrize("dtype", [np.float32, np.uint16, np.uint8])
def test_written_files_are_read_by_libtiff(tmp_path, compress, predictor, dtype):
    img, dem = synthetic_raster(173, 301, 5, hole=(10, 30, 40, 90))
    data = dem if dtype == np.float32 else (img * (60000 if dtype == np.uint16 else 255)).astype(dtype)
    path = str(tmp_path / "a.tif")
    G.write_geotiff(path, data, _geo_meta(), nodata=-32768.0, dtype=dtype, compress=compress, predictor=predictor)
    with Image.open(path) as im:
        back = np.array(im)
    assert back.shape == data.shape and np.array_equal(back.astype(np.float64), data.astype(np.float64))
    mine, meta = G.read_geotiff(path)
    assert mine.dtype == np.float32 and np.array_equal(mine, data.astype(np.float32))
    assert meta["nodata"] == -32768.0 and meta["geo"][33550][2] == _geo_meta()["geo"][33550][2]
    assert G.geotransform(meta) == (1000.0, 5.0, 0.0, -2000.0, 0.0, -5.0)


@pytest.mark.parametrize("compression", [None, "tiff_lzw", "tiff_adobe_deflate", "packbits"])
def test_libtiff_written_files_are_read(tmp_path, compression):
    _, dem = synthetic_raster(257, 190, 6)
    path = str(tmp_path / "b.tif")
    Image.fromarray(dem, mode="F").save(path, compression=compression)
    mine, meta = G.read_geotiff(path)
    assert np.array_equal(mine, dem) and meta["shape"] == dem.shape and meta["nodata"] is None
    # 16-bit with libtiff's own horizontal predictor
    u16 = (np.abs(dem) % 60000).astype(np.uint16)
    Image.fromarray(u16).save(path, compression="tiff_lzw", tiffinfo={317: 2})
    mine, _ = G.read_geotiff(path)
    assert np.array_equal(mine, u16.astype(np.float32))


def test_bigtiff_tiles_bigendian_and_band_selection(tmp_path):
    _, dem = synthetic_raster(70, 50, 7)
    p = str(tmp_path / "big.tif")
    G.write_geotiff(p, dem, _geo_meta(), nodata=-32768.0, bigtiff=True)
    assert open(p, "rb").read(4) == b"II+\x00"
    back, meta = G.read_geotiff(p)
    assert np.array_equal(back, dem) and meta["nodata"] == -32768.0
    # hand-built big-endian, tiled (16x16), 2-band chunky int16 file: band 2 = band 1 + 100
    rows, cols, tw = 40, 37, 16
    b1 = (np.arange(rows * cols).reshape(rows, cols) % 3000 - 1500).astype(np.int16)
    px = np.stack([b1, b1 + 100], -1)
    tiles, across, down = [], -(-cols // tw), -(-rows // tw)
    for ty in range(down):
        for tx in range(across):
            t = np.zeros((tw, tw, 2), ">i2")
            blk = px[ty * tw:(ty + 1) * tw, tx * tw:(tx + 1) * tw]
            t[:blk.shape[0], :blk.shape[1]] = blk
            tiles.append(t.tobytes())
    off = 8
    offs = []
    for t in tiles:
        offs.append(off)
        off += len(t)
    n = len(tiles)
    ent = [(256, 3, 1, struct.pack(">HH", cols, 0)), (257, 3, 1, struct.pack(">HH", rows, 0)),
           (258, 3, 2, struct.pack(">HH", 16, 16)), (259, 3, 1, struct.pack(">HH", 1, 0)),
           (262, 3, 1, struct.pack(">HH", 1, 0)), (277, 3, 1, struct.pack(">HH", 2, 0)),
           (284, 3, 1, struct.pack(">HH", 1, 0)), (322, 3, 1, struct.pack(">HH", tw, 0)),
           (323, 3, 1, struct.pack(">HH", tw, 0)), (324, 4, n, None), (325, 4, n, None),
           (339, 3, 2, struct.pack(">HH", 2, 2))]
    ifd_off = off
    extra_off = ifd_off + 2 + 12 * len(ent) + 4
    ifd = struct.pack(">H", len(ent))
    extra = b""
    for tag, typ, cnt, raw in ent:
        if raw is None:
            vals = offs if tag == 324 else [len(t) for t in tiles]
            ifd += struct.pack(">HHII", tag, typ, cnt, extra_off + len(extra))
            extra += struct.pack(f">{n}I", *vals)
        else:
            ifd += struct.pack(">HHI", tag, typ, cnt) + raw
    ifd += struct.pack(">I", 0)
    q = str(tmp_path / "tiled_be.tif")
    with open(q, "wb") as f:
        f.write(b"MM" + struct.pack(">HI", 42, ifd_off) + b"".join(tiles) + ifd + extra)
    # (PIL has no mode for 2-band int16, so this container is only read by the module under test)
    a1, meta = G.read_geotiff(q, band=1)
    a2, _ = G.read_geotiff(q, band=2)
    assert np.array_equal(a1, b1.astype(np.float32)) and np.array_equal(a2, a1 + 100) and meta["byteorder"] == ">"
    with pytest.raises(ValueError):
        G.read_geotiff(q, band=3)


def test_error_behaviour(tmp_path):
    p = str(tmp_path / "x.tif")
    open(p, "wb").write(b"not a tiff at all")
    with pytest.raises(ValueError):
        G.read_geotiff(p)
    with pytest.raises(ValueError):
        G.write_geotiff(p, np.zeros((2, 2, 2), np.float32))      # saveGTiff rejects rank != 2 (process_full_tiles.py:505-519)


def test_lzw_known_answer():
    # TIFF 6.0 section 13 example: the 7-byte string 7 7 7 8 8 7 7 6 6 encodes as the code stream
    # 256 7 258 8 8 258 6 6 257 (9-bit codes, MSB first)
    data = np.array([7, 7, 7, 8, 8, 7, 7, 6, 6], np.uint8)
    enc = G.lzw_encode(data)
    bits = "".join(f"{b:08b}" for b in enc)
    codes = [int(bits[i:i + 9], 2) for i in range(0, 9 * 9, 9)]
    assert codes == [256, 7, 258, 8, 8, 258, 6, 6, 257]
    assert np.array_equal(G.lzw_decode(enc, 9), data)
