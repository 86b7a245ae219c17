"""GPU (-m gpu): the patch-row-sharded ("halo") mode (moonsuperresolution_amd/halo.py) against its NumPy restatement
(oracle/tiler_ref.py::process_map_halo) and against the reference-identical tile mode.

What is compared with what, and to which tolerance:
  * identity model, one rank: the mean equals the TILE mode's mean bit for bit (every pixel sees the same patches in
    the same order; only the batch cut differs, which the identity model does not feel); std equals the oracle's
    textbook-West std bit for bit (S clamped at zero before the square root: finite wherever good == 1);
  * identity model, 2 and 3 simulated ranks (accumulate per rank, hand the boundary zones over in-process, finish):
    bit-exact against the oracle run with the same number of ranks; <= 1e-6 relative against one rank (pairwise
    instead of sequential combination in the zones);
  * HIP generator: <= 1e-3 of the DEM span against the oracle's halo mode driven by the oracle generator; the distance
    to the tile mode (other batch mates in SPADE's batch statistics) is printed — it is the mode's documented deviation.
"""
import numpy as np
import pytest
import torch

from oracle import tiler_ref
from tests.helpers import synthetic_raster

pytestmark = pytest.mark.gpu
NOVAL = -32768.0


def f32_identity(x, training=False):
    return np.asarray(x, np.float32)


def run_halo(d, img, dem, world, band_rows=None):
    """`world` ranks simulated on one GPU: phase 1 per rank, the zone hand-over done in-process, phase 2 per rank."""
    d.setImages(img, dem)
    states = [d.haloAccumulate(r, world, band_rows=band_rows) for r in range(world)]
    slabs = []
    for r, st in enumerate(states):
        from_down = states[r - 1]["send_up"] if r > 0 else None
        from_up = states[r + 1]["send_down"] if r < world - 1 else None
        assert from_down is None or tuple(from_down.shape) == (3, st["down_rows"], st["wp"])
        assert from_up is None or tuple(from_up.shape) == (3, st["up_rows"], st["wp"])
        slabs.append(d.haloFinish(st, from_down, from_up))
    assert slabs[0][1][0] == 0 and all(slabs[i][1][1] == slabs[i + 1][1][0] for i in range(world - 1))
    return d.cropHalo(slabs)


@pytest.mark.parametrize("S,stride,B,T,shape,hole", [
    (64, 16, 4, 128, (200, 330), (90, 110, 140, 170)),
    (64, 8, 16, 64, (150, 100), (20, 60, 40, 70)),
    (128, 32, 5, 256, (300, 280), None),
])
def test_identity_model_halo_mode(hip_lib, S, stride, B, T, shape, hole):
    from moonsuperresolution_amd import DSRConfig, HaloShardedSuperResolution
    img, dem = synthetic_raster(shape[0], shape[1], seed=S + stride, hole=hole)
    cfg = DSRConfig(image_size=S, stride=stride, batch_size=B, tile_size=T)
    d = HaloShardedSuperResolution(cfg, model=f32_identity)
    tile_mean, _, tile_good = d.processMap(img, dem)                     # the reference-identical mode
    one = run_halo(d, img, dem, 1)
    ref1 = tiler_ref.process_map_halo(img, dem, f32_identity, S, stride, B, T, NOVAL, world=1)
    assert np.array_equal(one[2], tile_good) and np.array_equal(one[0], tile_mean)
    for a, b in zip(one, ref1):
        assert np.array_equal(a, b, equal_nan=True)
    ok = one[2] == 1
    assert np.isfinite(one[1][ok]).all() and (one[1][ok] >= 0).all()     # S is clamped at zero: no NaN where good == 1
    for world in (2, 3):
        got = run_halo(d, img, dem, world)
        ref = tiler_ref.process_map_halo(img, dem, f32_identity, S, stride, B, T, NOVAL, world=world)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b, equal_nan=True), world
        assert np.array_equal(got[2], one[2])
        scale = float(np.abs(one[0][ok]).max())
        assert float(np.abs(got[0][ok] - one[0][ok]).max()) <= 1e-6 * scale
    # banded accumulation (the sliding window over patch rows): the rank's patch rows in bands of 1, 2 and 3 rows, each
    # band continued in place from what the earlier bands left -> the same bits as all rows at once, for 1 and 3 ranks
    ys, _ = d.patchGrid()
    for world, band_rows in ((1, 1), (1, 3), (3, 1), (3, 2)):
        assert -(-(len(ys) // world) // band_rows) >= 3                    # at least three bands per rank
        got = run_halo(d, img, dem, world, band_rows=band_rows)
        ref = one if world == 1 else tiler_ref.process_map_halo(img, dem, f32_identity, S, stride, B, T, NOVAL, world=world)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b, equal_nan=True), (world, band_rows)
    d.close()


def test_halo_overlapped_finish_and_max_rows(hip_lib):
    """haloFinish(exchange=wait) — interior rows finalised before the neighbours' slabs are taken — gives the bits of the
    blocking order; max_rows stops after that many patch rows (the rest of the rank's accumulator stays empty)."""
    from moonsuperresolution_amd import DSRConfig, HaloShardedSuperResolution
    img, dem = synthetic_raster(300, 200, 5)
    d = HaloShardedSuperResolution(DSRConfig(image_size=64, stride=16, batch_size=4, tile_size=128), model=f32_identity)
    d.setImages(img, dem)
    states = [d.haloAccumulate(r, 3, band_rows=2) for r in range(3)]
    st = states[1]
    a = d.haloFinish(st, states[0]["send_up"], states[2]["send_down"])
    b = d.haloFinish(st, exchange=lambda: (states[0]["send_up"], states[2]["send_down"]))
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    full = d.haloAccumulate(0, 1, band_rows=2)
    part = d.haloAccumulate(0, 1, band_rows=2, max_rows=5)
    ys, _ = d.patchGrid()
    p = 64 // 16
    reach = ys[4] + 64 - p                                   # canvas rows the first five patch rows reach
    untouched_from = ys[5] + p                               # rows below this line got every patch they will ever get
    lo = full["lo"]                                          # accumulator row 0 is canvas row lo
    assert torch.equal(part["acc"][:, :untouched_from - lo], full["acc"][:, :untouched_from - lo])
    assert float(part["acc"][0, reach - lo:].abs().max()) == 0.0 and float(full["acc"][0, reach - lo:].abs().max()) > 0.0
    d.close()


def test_halo_mode_rejects_a_stride_that_does_not_divide(hip_lib):
    """msr_stitch_partial bins patches by origin / stride relative to each T x T block: with S % s or T % s the
    patches would be dropped without an error, so the mode refuses such a configuration (the tile mode accepts it)."""
    from moonsuperresolution_amd import DSRConfig, HaloShardedSuperResolution
    img, dem = synthetic_raster(150, 140, 3)
    for S, stride, T in ((64, 24, 128), (64, 16, 120)):
        d = HaloShardedSuperResolution(DSRConfig(image_size=S, stride=stride, batch_size=4, tile_size=T), model=f32_identity)
        d.setImages(img, dem)
        with pytest.raises(ValueError, match="stride"):
            d.haloAccumulate(0, 1)
        d.close()


def test_generator_halo_mode_vs_oracle_and_tile_mode(hip_lib):
    from moonsuperresolution_amd import DSRConfig, Generator, HaloShardedSuperResolution, make_weights
    from oracle import generator_ref
    w = make_weights("gaugan_no_kl", 64, seed=1234, bias_scale=0.05)
    img, dem = synthetic_raster(150, 140, 9)
    cfg = DSRConfig(image_size=64, stride=16, batch_size=4, tile_size=128)
    gen = Generator(64, 4, variant="gaugan_no_kl", weights=w)
    d = HaloShardedSuperResolution(cfg, model=gen)
    tile_mean, tile_std, tile_good = d.processMap(img, dem)
    got = run_halo(d, img, dem, 2)
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    ref = tiler_ref.process_map_halo(
        img, dem, lambda x, training=False: generator_ref.spade_call(x, wt, "gaugan_no_kl", dtype=torch.float32),
        64, 16, 4, 128, NOVAL, world=2)
    assert np.array_equal(got[2], ref[2]) and np.array_equal(got[2], tile_good) and got[2].any()
    ok = got[2] == 1
    span = float(dem.max() - dem.min())
    assert np.abs(got[0][ok] - ref[0][ok]).max() <= 1e-3 * span
    assert np.nanmax(np.abs(got[1][ok] - ref[1][ok])) <= 1e-3 * span
    dev = float(np.abs(got[0][ok] - tile_mean[ok]).max() / span)
    print(f"halo mode vs tile mode (other batch composition): max |mean difference| = {dev:.3e} of the DEM span")
    assert dev < 0.5           # same picture, different batch statistics: a loose sanity bound, not a parity claim
    d.close(); gen.close()
