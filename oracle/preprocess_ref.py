"""ORACLE (test infrastructure, never shipped, never on the product path).

NumPy / SciPy restatement of the reference's raster pre-processing (``DEMSuperResolution.preprocess`` and its
helpers, process_full_tiles.py:184-244): nodata in-filling and the synthesis of the low-resolution DEM
(x1/4 INTER_AREA -> fill -> x1/4 INTER_AREA -> INTER_CUBIC back to full size).

PARITY UNPINNED for the two resamplers: the reference calls ``cv2.resize`` (opencv-python, not installed here, no
vendored copy) and holds no fixture of its output, so ``resize_area`` / ``resize_cubic`` restate OpenCV's PUBLISHED
algorithm (modules/imgproc/src/resize.cpp of the 4.x line): destination size and scale as ``cv::resize`` derives
them, the integer-scale "area fast" path, Keys' cubic with A = -0.75 evaluated in float32, pixel-centre mapping
``(d + 0.5) * scale - 0.5``, replicated borders, horizontal pass then vertical pass.  Bit-level agreement with a
real OpenCV build is not claimed.  The in-filling uses the same SciPy routine the reference calls
(``scipy.interpolate.griddata(method="cubic")``), and ``scipy.ndimage.label`` with the 8-connected structure in
place of ``cv2.connectedComponents`` (default connectivity 8): only region membership and sizes are used, which do
not depend on label numbering.

Reference quirks that are reproduced, not fixed:
* ``np.unique(blobs[1], return_counts=True)`` counts the background label 0 (= the VALID pixels) together with the
  holes, so "the smallest region is larger than max_fill_area" and the keep-loop both see it (:196-212);
* ``preprocess`` stores the filled ortho in ``self.image`` while the rest of the driver reads ``self.img`` (:227
  vs :261): the filled ortho is returned here but unused by the driver;
* ``cv2.resize(dem_rs, self.dem_shape, ...)`` passes (rows, cols) where OpenCV expects (width, height) (:241): the
  result has shape (cols, rows), i.e. the reference's pre-processing only works for square rasters.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

F32 = np.float32


def cv_round(v: float) -> int:
    """cvRound / saturate_cast<int>(double): round half to even."""
    return int(np.rint(v))


def resize_area(src: np.ndarray, fx: float = 0.25, fy: float = 0.25) -> np.ndarray:
    """cv2.resize(src, (0,0), fx, fy, INTER_AREA) for float32 and an integer reduction factor (1/fx, 1/fy integers:
    the "area fast" path).  Every destination pixel is the float32 sum of its iscale_y x iscale_x source block
    (rows outer, four columns at a time as ``sum += S[k] + S[k+1] + S[k+2] + S[k+3]``) times float32(1/area);
    blocks that stick out of the source (size not divisible) average the pixels that exist.  NaNs propagate."""
    src = np.ascontiguousarray(src, F32)
    h, w = src.shape
    dw, dh = cv_round(w * fx), cv_round(h * fy)
    sx, sy = int(round(1.0 / fx)), int(round(1.0 / fy))
    if abs(1.0 / fx - sx) > 1e-12 or abs(1.0 / fy - sy) > 1e-12:
        raise ValueError("only integer reduction factors are restated (the reference uses 0.25)")
    out = np.empty((dh, dw), F32)
    full_w, full_h = min(dw, w // sx), min(dh, h // sy)
    scale = F32(1.0 / (sx * sy))
    # full blocks, OpenCV's summation order
    blk = src[:full_h * sy, :full_w * sx].reshape(full_h, sy, full_w, sx)
    total = np.zeros((full_h, full_w), F32)
    for r in range(sy):
        k = 0
        while k + 4 <= sx:
            part = ((blk[:, r, :, k] + blk[:, r, :, k + 1]) + blk[:, r, :, k + 2]) + blk[:, r, :, k + 3]
            total = total + part
            k += 4
        while k < sx:
            total = total + blk[:, r, :, k]
            k += 1
    out[:full_h, :full_w] = total * scale
    # partial blocks on the right / bottom edge: plain mean of the pixels that exist (float32 running sum)
    for dy in range(dh):
        for dx in range(dw):
            if dy < full_h and dx < full_w:
                continue
            y0, x0 = dy * sy, dx * sx
            if y0 >= h or x0 >= w:
                out[dy, dx] = 0.0
                continue
            s, n = F32(0), 0
            for yy in range(y0, min(y0 + sy, h)):
                for xx in range(x0, min(x0 + sx, w)):
                    s = F32(s + src[yy, xx])
                    n += 1
            out[dy, dx] = F32(s / F32(n))
    return out


def cubic_coeffs(t: np.ndarray) -> np.ndarray:
    """interpolateCubic (imgproc/src/precomp.hpp), A = -0.75, float32 arithmetic; t in [0, 1)."""
    A = F32(-0.75)
    t = t.astype(F32)
    one = F32(1)
    c0 = ((A * (t + one) - F32(5) * A) * (t + one) + F32(8) * A) * (t + one) - F32(4) * A
    c1 = ((A + F32(2)) * t - (A + F32(3))) * t * t + one
    u = one - t
    c2 = ((A + F32(2)) * u - (A + F32(3))) * u * u + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], -1).astype(F32)


def _cubic_axis(n_src: int, n_dst: int) -> Tuple[np.ndarray, np.ndarray]:
    """Source tap indices [n_dst, 4] (clamped = replicated border) and float32 weights [n_dst, 4] of one axis."""
    scale = n_src / n_dst                                   # double, as cv::resize computes 1 / inv_scale
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(F32)               # float fx = (float)((dx + 0.5) * scale_x - 0.5)
    s = np.floor(f).astype(np.int64)
    t = (f - s.astype(F32)).astype(F32)
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
    return idx, cubic_coeffs(t)


def resize_cubic(src: np.ndarray, dsize_wh: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(src, (width, height), interpolation=INTER_CUBIC) for float32: horizontal pass into float32 rows
    (``S[-1]*a0 + S[0]*a1 + S[1]*a2 + S[2]*a3``, left to right), then the vertical pass over four such rows
    (``S0*b0 + S1*b1 + S2*b2 + S3*b3``).  NaNs propagate through both passes."""
    src = np.ascontiguousarray(src, F32)
    h, w = src.shape
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    xi, xa = _cubic_axis(w, dw)
    yi, yb = _cubic_axis(h, dh)
    rows = (((src[:, xi[:, 0]] * xa[:, 0]) + src[:, xi[:, 1]] * xa[:, 1]) + src[:, xi[:, 2]] * xa[:, 2]) \
        + src[:, xi[:, 3]] * xa[:, 3]                       # [h, dw] float32
    rows = rows.astype(F32)
    out = np.empty((dh, dw), F32)
    step = max(1, (1 << 24) // max(dw, 1))
    for y0 in range(0, dh, step):                           # blocked: the full-size result can be several GB
        sl = slice(y0, min(dh, y0 + step))
        b = yb[sl]
        r = yi[sl]
        out[sl] = (((rows[r[:, 0]] * b[:, 0:1]) + rows[r[:, 1]] * b[:, 1:2]) + rows[r[:, 2]] * b[:, 2:3]) \
            + rows[r[:, 3]] * b[:, 3:4]
    return out


def interpolate_missing_values(data: np.ndarray, no_value: float, max_fill_area: int = 256) -> np.ndarray:
    """interpolateMissingValues, process_full_tiles.py:184-212 (in place on ``data``, which it also returns)."""
    from scipy import interpolate, ndimage
    bad = data <= no_value
    n_bad = int(bad.sum())
    if n_bad == 0 or n_bad == bad.size:
        return data
    lab, n_lab = ndimage.label(bad, structure=np.ones((3, 3), dtype=bool))   # cv2 default: 8-connected
    # the reference takes np.unique over the whole label image: label 0 (the valid pixels) is one of the "regions"
    census = {k: int((lab == k).sum()) for k in range(n_lab + 1)}
    if min(census.values()) > max_fill_area:
        return data
    yy, xx = np.indices(data.shape)
    good = ~bad
    cubic = interpolate.griddata((xx[good], yy[good]), data[good], (xx, yy), method="cubic")
    for k, size in census.items():
        if size < max_fill_area:
            sel = lab == k
            data[sel] = cubic[sel]
    return data


def fill_nan(image: np.ndarray, no_value: float, tile_size: int = 1024, border: int = 128,
             max_fill_area: int = 256) -> np.ndarray:
    """fillNan, process_full_tiles.py:214-224: overlapping tiles, only each tile's interior is written back."""
    result = np.array(image, copy=True)
    rows, cols = image.shape
    pitch = tile_size - border * 2
    for y0 in range(0, rows, pitch):
        y1 = min(y0 + tile_size - border, rows - border)
        for x0 in range(0, cols, pitch):
            x1 = min(x0 + tile_size - border, cols - border)
            patch = np.array(image[y0:y0 + tile_size, x0:x0 + tile_size], copy=True)
            patch = interpolate_missing_values(patch, no_value, max_fill_area=max_fill_area)
            result[y0 + border:y1, x0 + border:x1] = patch[border:-border, border:-border]
    return result


def preprocess(img: np.ndarray, dem: np.ndarray, no_value: float) -> Tuple[np.ndarray, np.ndarray]:
    """preprocess, process_full_tiles.py:226-244.  Returns (filled ortho — stored but never used by the reference —,
    low-resolution DEM resampled back to ``dem.shape`` interpreted as (width, height), i.e. shape (cols, rows))."""
    image = fill_nan(np.asarray(img, F32), no_value, tile_size=1024, border=128, max_fill_area=8)
    dem_rs = np.array(dem, F32, copy=True)
    dem_rs[dem_rs <= no_value] = np.nan
    dem_rs = resize_area(dem_rs, 0.25, 0.25)
    dem_rs[np.isnan(dem_rs)] = no_value
    dem_rs = fill_nan(dem_rs, no_value, tile_size=256, border=32, max_fill_area=24)
    dem_rs[dem_rs <= no_value] = np.nan
    dem_rs = resize_area(dem_rs, 0.25, 0.25)
    dem_rs = resize_cubic(dem_rs, (dem.shape[0], dem.shape[1]))     # (rows, cols) passed as (width, height): :241
    dem_rs[np.isnan(dem_rs)] = no_value
    return image, dem_rs
