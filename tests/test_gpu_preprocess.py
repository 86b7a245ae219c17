"""GPU tests of the pre-processing row: msr_resize_area / msr_resize_cubic through the C ABI are
bit-exact against oracle/preprocess_ref.py, and DEMSuperResolution.preprocess equals the oracle's preprocess
(the in-filling is the same SciPy call on both sides)."""
import numpy as np
import pytest
import torch

from oracle import preprocess_ref as pr

pytestmark = pytest.mark.gpu
NOVAL = -32768.0


@pytest.fixture(scope="module")
def ctx(hip_lib):
    from moonsuperresolution_amd import ops
    return ops.OpContext()


def eq_nan(a, b):
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b))


@pytest.mark.parametrize("shape", [(64, 64), (50, 75), (10, 11), (6, 6), (1030, 517)])
def test_resize_area_bit_exact(ctx, shape):
    from moonsuperresolution_amd import preprocess as pp
    rng = np.random.default_rng(shape[0])
    src = (rng.normal(size=shape) * 1000).astype(np.float32)
    src[rng.uniform(size=shape) < 0.01] = np.nan
    out = pp.resize_area(ctx.lib, ctx.h, torch.from_numpy(src).cuda(), 4).cpu().numpy()
    assert eq_nan(out, pr.resize_area(src, 0.25, 0.25))


@pytest.mark.parametrize("shape,dsize", [((7, 9), (9, 7)), ((16, 12), (192, 256)), ((5, 6), (24, 20)), ((33, 65), (1000, 517)),
                                         ((40, 40), (13, 17))])
def test_resize_cubic_bit_exact(ctx, shape, dsize):
    from moonsuperresolution_amd import preprocess as pp
    rng = np.random.default_rng(dsize[0])
    src = (rng.normal(size=shape) * 100 - 2000).astype(np.float32)
    if shape[0] > 8:
        src[3, 4] = np.nan
    out = pp.resize_cubic(ctx.lib, ctx.h, torch.from_numpy(src).cuda(), dsize).cpu().numpy()
    assert eq_nan(out, pr.resize_cubic(src, dsize))


def surface(h, w):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    return (-2000.0 + 0.3 * xx + 0.2 * yy + 5.0 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.float32)


def test_preprocess_matches_oracle_and_driver_quirks():
    from moonsuperresolution_amd import DEMSuperResolution, DSRConfig
    dem = surface(1536, 1536)
    dem[700:702, 900:903] = NOVAL
    dem[300:340, 300:340] = NOVAL          # too large to in-fill: survives as no_value
    img = np.random.default_rng(3).uniform(0, 1, dem.shape).astype(np.float32)   # no holes: a 1024^2 griddata is slow
    d = DEMSuperResolution(DSRConfig(image_size=64, stride=32, batch_size=4, tile_size=128), model=lambda x, training=False: x)
    d.setImages(img, dem)
    d.preprocess()
    ref_image, ref_dem = pr.preprocess(img, dem, NOVAL)
    assert np.array_equal(d.dem, ref_dem) and d.dem.dtype == np.float32
    assert np.array_equal(d.image, ref_image)
    assert d.img is img or np.array_equal(d.img, img)        # the driver keeps reading the UN-filled ortho (:227 vs :261)
    assert (d.dem == NOVAL).any()
    # non-square: (rows, cols) handed over as (width, height) (:241) -> transposed shape, as in the reference
    d.setImages(img[:512, :1024], dem[:512, :1024])
    d.preprocess()
    assert d.dem.shape == (1024, 512)
    d.setImages(img[:512, :1024], dem[:512, :1024])
    d.preprocess(swap_dsize=False)
    assert d.dem.shape == (512, 1024)
    with pytest.raises(ValueError):
        e = DEMSuperResolution(DSRConfig(image_size=64, stride=32, batch_size=4, tile_size=128), model=lambda x, training=False: x)
        e.preprocess()
    d.close()
