"""Shared seeded input builders for tests and tests/golden/make_golden.py."""
import numpy as np


def rel_linf(a, b):
    """Relative L-infinity error used for every floating-point parity statement: max|a-b| / max|b|."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def stitch_inputs(S=64, s=16, T=128, seed=5, drop=0.15):
    """Random generator outputs for one tile in generation order: keys [n,2] (x,y), pred [n,S,S] (raw model
    output, i.e. before the +0.5 of process_full_tiles.py:340), dem (min,max) [n,2]."""
    rng = np.random.default_rng(seed)
    n_side = len(range(0, T + S - s, s))
    keys, pred, mm = [], [], []
    for iy in range(n_side):
        for ix in range(n_side):
            if rng.uniform() < drop:
                continue   # invalid patches are simply absent (process_full_tiles.py:456-457)
            keys.append((ix * s, iy * s))
            pred.append(rng.uniform(-0.6, 0.6, (S, S)).astype(np.float32))
            lo = np.float32(rng.uniform(-3000, -2000))
            mm.append((lo, np.float32(lo + rng.uniform(5, 400))))
    return np.array(keys, np.int32), np.stack(pred), np.array(mm, np.float32)


def synthetic_raster(h, w, seed=0, hole=None, no_value=-32768.0):
    """Smooth (ortho, DEM) rasters: ortho in [0,1], DEM in [-3000,-1000] m, optional nodata rectangle."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    dem = (-2000 + 600 * np.sin(xx / 37.0) * np.cos(yy / 53.0) + 300 * np.sin((xx + 2 * yy) / 91.0)
           + 5 * rng.standard_normal((h, w)))
    img = 0.5 + 0.3 * np.cos(xx / 23.0) * np.sin(yy / 31.0) + 0.05 * rng.standard_normal((h, w))
    dem = dem.astype(np.float32)
    img = np.clip(img, 0, 1).astype(np.float32)
    if hole is not None:
        y0, y1, x0, x1 = hole
        dem[y0:y1, x0:x1] = no_value
    return img, dem
