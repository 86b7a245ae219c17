"""Raster pre-processing of the inference driver (process_full_tiles.py:184-244), SURVEY.md 8f rank 3.

``preprocess`` = nodata in-filling of the ortho-image (result stored, never used — reference quirk), then the
synthesis of the low-resolution DEM: x1/4 INTER_AREA -> in-filling -> x1/4 INTER_AREA -> INTER_CUBIC back to full
size.  The two resamplers run on the GPU through the C ABI (``msr_resize_area`` / ``msr_resize_cubic``; the
full-resolution raster is read and written once); the in-filling stays on the host exactly like the reference: the
same ``scipy.interpolate.griddata(method="cubic")`` call on each tile that has holes, with ``scipy.ndimage.label``
(8-connected) in place of ``cv2.connectedComponents``.

OpenCV is not available here and the reference holds no fixture of ``cv2.resize`` output, so the resamplers follow
OpenCV's published algorithm (see oracle/preprocess_ref.py, against which the kernels are bit-exact): parity with a
real OpenCV build is unpinned.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np
import torch

from . import _lib


def _cv_round(v: float) -> int:
    return int(np.rint(v))


def resize_area(lib, handle, src: torch.Tensor, factor: int = 4) -> torch.Tensor:
    """cv2.resize(src, (0,0), fx=1/factor, fy=1/factor, interpolation=cv2.INTER_AREA) on a CUDA float32 raster."""
    if src.dim() != 2 or src.dtype != torch.float32 or not src.is_cuda:
        raise ValueError("resize_area expects a 2-D float32 CUDA tensor")
    src = src.contiguous()
    h, w = src.shape
    dh, dw = _cv_round(h / factor), _cv_round(w / factor)
    dst = torch.empty((dh, dw), dtype=torch.float32, device=src.device)
    rc = lib.msr_resize_area(handle, src.data_ptr(), h, w, factor, dst.data_ptr(), dh, dw,
                             torch.cuda.current_stream(src.device).cuda_stream)
    _lib.raise_for(lib, handle, rc, "msr_resize_area")
    return dst


def resize_cubic(lib, handle, src: torch.Tensor, dsize_wh: Tuple[int, int]) -> torch.Tensor:
    """cv2.resize(src, (width, height), interpolation=cv2.INTER_CUBIC) on a CUDA float32 raster."""
    if src.dim() != 2 or src.dtype != torch.float32 or not src.is_cuda:
        raise ValueError("resize_cubic expects a 2-D float32 CUDA tensor")
    src = src.contiguous()
    h, w = src.shape
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    dst = torch.empty((dh, dw), dtype=torch.float32, device=src.device)
    rc = lib.msr_resize_cubic(handle, src.data_ptr(), h, w, dst.data_ptr(), dh, dw,
                              torch.cuda.current_stream(src.device).cuda_stream)
    _lib.raise_for(lib, handle, rc, "msr_resize_cubic")
    return dst


def interpolateMissingValues(data: np.ndarray, no_value: float, max_fill_area: int = 256) -> np.ndarray:
    """process_full_tiles.py:184-212 (in place; also returned).  A cubic ``griddata`` surface through the valid pixels of
    the tile; only 8-connected hole regions smaller than ``max_fill_area`` take its values.  Like the reference, the
    region census includes label 0 — the VALID pixels — so it takes part in the "smallest region too large" test and
    in the size filter."""
    from scipy import interpolate, ndimage
    holes = data <= no_value
    if not holes.any() or holes.all():          # nothing to fill / nothing to interpolate from
        return data
    labels, _ = ndimage.label(holes, structure=np.ones((3, 3), dtype=bool))
    sizes = np.bincount(labels.ravel())         # sizes[0] = number of valid pixels, sizes[k] = pixels of hole k
    if sizes.min() > max_fill_area:             # every region (the valid area included) is too large
        return data
    rows, cols = np.nonzero(~holes)             # row-major, the order of the reference's boolean indexing
    grid_r, grid_c = np.mgrid[0:data.shape[0], 0:data.shape[1]]
    surface = interpolate.griddata((cols, rows), data[rows, cols], (grid_c, grid_r), method="cubic")
    take = (sizes < max_fill_area)[labels]
    data[take] = surface[take]
    return data


def fillNan(image: np.ndarray, no_value: float, tile_size: int = 1024, border: int = 128,
            max_fill_area: int = 256) -> np.ndarray:
    """process_full_tiles.py:214-224: tiles of ``tile_size`` every ``tile_size - 2*border`` pixels, each filled on its
    own; only the part of a tile inside its ``border`` frame is written back, clipped ``border`` short of the raster."""
    h, w = image.shape
    out = image.copy()
    step = tile_size - 2 * border
    for top in range(0, h, step):
        bottom = min(top + tile_size - border, h - border)
        for left in range(0, w, step):
            right = min(left + tile_size - border, w - border)
            tile = interpolateMissingValues(image[top:top + tile_size, left:left + tile_size].copy(), no_value,
                                            max_fill_area=max_fill_area)
            out[top + border:bottom, left + border:right] = tile[border:-border, border:-border]
    return out


def preprocess(lib, handle, device, img: np.ndarray, dem: np.ndarray, no_value: float,
               swap_dsize: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """process_full_tiles.py:226-244.  Returns (filled ortho, low-resolution DEM at full size), both host float32.

    ``swap_dsize=True`` reproduces :241, where ``self.dem_shape`` = (rows, cols) is handed to ``cv2.resize`` as
    (width, height): the result then has shape (cols, rows), so — like the reference — only square rasters survive
    the later ``padInputs``.  ``swap_dsize=False`` resizes to (rows, cols) proper."""
    img = np.asarray(img, np.float32)
    dem = np.asarray(dem, np.float32)
    image = fillNan(img, no_value, tile_size=1024, border=128, max_fill_area=8)
    with torch.cuda.device(device):
        d = torch.from_numpy(np.ascontiguousarray(dem)).to(device)
        d = torch.where(d <= no_value, torch.full_like(d, float("nan")), d)      # keep OpenCV from averaging no_values
        d4 = resize_area(lib, handle, d, 4)
        del d
        d4 = torch.where(torch.isnan(d4), torch.full_like(d4, no_value), d4)
        h4 = fillNan(d4.cpu().numpy(), no_value, tile_size=256, border=32, max_fill_area=24)
        d4 = torch.from_numpy(h4).to(device)
        d4 = torch.where(d4 <= no_value, torch.full_like(d4, float("nan")), d4)
        d16 = resize_area(lib, handle, d4, 4)
        dsize = (dem.shape[0], dem.shape[1]) if swap_dsize else (dem.shape[1], dem.shape[0])
        up = resize_cubic(lib, handle, d16, dsize)
        up = torch.where(torch.isnan(up), torch.full_like(up, no_value), up)
        out = up.cpu().numpy()
    return image, out
