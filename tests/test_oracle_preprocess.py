"""CPU tests of oracle/preprocess_ref.py — the restatement of process_full_tiles.py:184-244 (nodata in-filling,
low-resolution DEM synthesis).  The resamplers restate OpenCV's published INTER_AREA / INTER_CUBIC algorithm
("parity unpinned": no OpenCV here, no fixture in the reference); these known-answer tests pin the semantics the
restatement claims: destination size by cvRound, block means with NaN propagation, Keys' A = -0.75 weights, the
pixel-centre mapping and the replicated border."""
import numpy as np
import pytest

from oracle import preprocess_ref as pr

NOVAL = -32768.0


def test_area_block_means_and_nan():
    src = np.arange(64, dtype=np.float32).reshape(8, 8)
    out = pr.resize_area(src)
    assert out.shape == (2, 2)
    assert np.array_equal(out, np.array([[13.5, 17.5], [45.5, 49.5]], np.float32))
    src[5, 6] = np.nan
    out = pr.resize_area(src)
    assert np.isnan(out[1, 1]) and np.isfinite(out[0, 0]) and np.isfinite(out[1, 0])


def test_area_destination_size_is_cvround_and_edges_average_what_exists():
    src = np.ones((10, 11), np.float32)
    src[:, 8:] = 5.0
    out = pr.resize_area(src)                       # rows: cvRound(2.5) = 2 (half to even); cols: cvRound(2.75) = 3
    assert out.shape == (2, 3)
    assert np.array_equal(out[:, :2], np.ones((2, 2), np.float32))
    assert np.array_equal(out[:, 2], np.full(2, 5.0, np.float32))      # 3 existing columns of the partial block
    assert pr.resize_area(np.ones((6, 6), np.float32)).shape == (2, 2)  # cvRound(1.5) = 2: partial bottom/right blocks


def test_cubic_weights_known_values():
    c = pr.cubic_coeffs(np.array([0.0, 0.5], np.float32))
    assert np.array_equal(c[0], np.array([0, 1, 0, 0], np.float32))
    assert np.allclose(c[1], [-0.09375, 0.59375, 0.59375, -0.09375], atol=1e-7)   # Keys, A = -0.75
    assert np.allclose(pr.cubic_coeffs(np.linspace(0, 0.99, 50).astype(np.float32)).sum(-1), 1.0, atol=1e-6)


def test_cubic_identity_constant_and_border():
    rng = np.random.default_rng(0)
    src = rng.normal(size=(7, 9)).astype(np.float32)
    assert np.array_equal(pr.resize_cubic(src, (9, 7)), src)                      # same size: t = 0 everywhere
    assert np.allclose(pr.resize_cubic(np.full((5, 6), 3.25, np.float32), (24, 20)), 3.25, atol=1e-6)
    # 2x up-sampling of a row: destination 0 maps to source -0.25 -> taps (-2..1) clamp to (0,0,0,1), t = 0.75
    row = np.array([[1.0, 2.0, 4.0, 8.0]], np.float32)
    up = pr.resize_cubic(row, (8, 1))
    w = pr.cubic_coeffs(np.array([0.75], np.float32))[0]
    assert up.shape == (1, 8)
    assert np.isclose(up[0, 0], w[0] * 1 + w[1] * 1 + w[2] * 1 + w[3] * 2)
    # destination 3 maps to source 1.25: taps 0..3, t = 0.25
    w = pr.cubic_coeffs(np.array([0.25], np.float32))[0]
    assert np.isclose(up[0, 3], w[0] * 1 + w[1] * 2 + w[2] * 4 + w[3] * 8)
    nan_src = src.copy()
    nan_src[3, 4] = np.nan
    out = pr.resize_cubic(nan_src, (18, 14))
    assert np.isnan(out).any() and np.isfinite(out[0, 0])


def surface(h, w):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    return (100.0 + 0.3 * xx + 0.2 * yy + 5.0 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.float32)


def test_interpolate_missing_values_fills_small_holes_only():
    truth = surface(64, 64)
    data = truth.copy()
    data[20:22, 30:33] = NOVAL            # 6 pixels: filled
    data[40:50, 10:22] = NOVAL            # 120 pixels: left alone with max_fill_area = 24
    out = pr.interpolate_missing_values(data.copy(), NOVAL, max_fill_area=24)
    assert np.allclose(out[20:22, 30:33], truth[20:22, 30:33], atol=0.05)
    assert (out[40:50, 10:22] == NOVAL).all()
    untouched = np.ones_like(data, bool)
    untouched[20:22, 30:33] = False
    assert np.array_equal(out[untouched], data[untouched])
    # nothing missing / everything missing / only areas above the limit: returned as is (:188-201)
    assert np.array_equal(pr.interpolate_missing_values(truth.copy(), NOVAL), truth)
    allbad = np.full((8, 8), NOVAL, np.float32)
    assert np.array_equal(pr.interpolate_missing_values(allbad.copy(), NOVAL), allbad)


def test_background_label_is_counted_like_a_region():
    """np.unique(labels) includes label 0 = the valid pixels (process_full_tiles.py:197): a tile whose VALID area is
    smaller than max_fill_area passes the 'too large' test even though its hole is big, and the hole stays because
    its own count is not below the limit."""
    data = np.full((12, 12), NOVAL, np.float32)
    data[0:2, 0:3] = 7.0                  # 6 valid pixels, 138 missing in one region
    data[0, 0] = 1.0
    out = pr.interpolate_missing_values(data.copy(), NOVAL, max_fill_area=24)
    assert (out[5:, 5:] == NOVAL).all()


def test_fill_nan_writes_tile_interiors_only():
    truth = surface(96, 96)
    img = truth.copy()
    img[2, 3] = NOVAL                     # inside the 8-pixel frame that no tile interior covers: never filled
    img[40, 41] = NOVAL
    out = pr.fill_nan(img, NOVAL, tile_size=32, border=8, max_fill_area=8)
    assert out[2, 3] == NOVAL
    assert abs(out[40, 41] - truth[40, 41]) < 0.05
    assert out.shape == img.shape


def test_product_infilling_equals_oracle():
    """moonsuperresolution_amd.preprocess keeps the in-filling on the host (SciPy, as the reference): it and the oracle
    are written independently and must agree exactly, quirks included."""
    from moonsuperresolution_amd import preprocess as pp
    rng = np.random.default_rng(5)
    img = surface(120, 150)
    for _ in range(12):
        y, x = int(rng.integers(0, 118)), int(rng.integers(0, 148))
        img[y:y + int(rng.integers(1, 4)), x:x + int(rng.integers(1, 4))] = NOVAL
    img[60:80, 90:120] = NOVAL
    for kw in (dict(tile_size=32, border=8, max_fill_area=8), dict(tile_size=64, border=16, max_fill_area=24)):
        assert np.array_equal(pp.fillNan(img.copy(), NOVAL, **kw), pr.fill_nan(img.copy(), NOVAL, **kw))
    tiny = np.full((12, 12), NOVAL, np.float32)
    tiny[0:2, 0:3] = 7.0
    assert np.array_equal(pp.interpolateMissingValues(tiny.copy(), NOVAL, 24), pr.interpolate_missing_values(tiny.copy(), NOVAL, 24))


def test_preprocess_square_raster():
    truth = surface(2048, 2048)
    dem = truth.copy()
    dem[1000:1002, 500:503] = NOVAL        # one quarter-resolution pixel goes missing and is in-filled there
    img = np.random.default_rng(1).uniform(0, 1, (2048, 2048)).astype(np.float32)
    image, low = pr.preprocess(img, dem, NOVAL)
    assert image.shape == img.shape and low.shape == (2048, 2048) and low.dtype == np.float32
    assert (low > NOVAL).all()
    assert np.abs(low - truth)[64:-64, 64:-64].max() < 2.5          # a 16x smoothed copy of a smooth surface
    # a raster smaller than one in-filling tile interior is never in-filled (fillNan writes [border:-border] of
    # each tile only, :224): the hole survives both reductions and comes back as no_value
    small = surface(256, 256)
    small[100:102, 50:53] = NOVAL
    _, low_s = pr.preprocess(img[:256, :256], small, NOVAL)
    assert (low_s == NOVAL).any()
    # the reference hands (rows, cols) to cv2.resize as (width, height): a non-square raster comes back transposed
    _, low2 = pr.preprocess(img[:128, :256], truth[:128, :256].copy(), NOVAL)
    assert low2.shape == (256, 128)
