"""GPU (-m gpu): tiler / stitcher kernels through the C ABI, bit-exact against the NumPy oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import tiler_ref
from tests.helpers import rel_linf, stitch_inputs, synthetic_raster

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
NOVAL = -32768.0


def f32_identity(x, training=False):
    """The reference's identity self-check, returning float32 like a real model does."""
    return np.asarray(x, np.float32)


@pytest.fixture(scope="module")
def dsr(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from moonsuperresolution_amd import DEMSuperResolution, DSRConfig
    return DEMSuperResolution, DSRConfig


def test_stitch_bit_exact_vs_golden_and_oracle(dsr):
    DEMSuperResolution, DSRConfig = dsr
    g = np.load(os.path.join(GOLD, "stitch_small.npz"))
    keys, pred, mm = stitch_inputs()
    for as_impl, km, ks in ((True, "mean", "std"), (False, "mean_textbook", "std_textbook")):
        d = DEMSuperResolution(DSRConfig(image_size=64, stride=16, batch_size=4, tile_size=128), as_implemented=as_impl)
        mean, std, good = d.rebuildTile(torch.from_numpy(pred).cuda(), torch.from_numpy(keys).cuda(),
                                        torch.from_numpy(mm).cuda())
        assert np.array_equal(mean.cpu().numpy(), g[km])          # bit-exact, float32
        assert np.array_equal(std.cpu().numpy(), g[ks])
        assert np.array_equal(good.cpu().numpy(), g["good"])
        d.close()


def test_patch_stats_and_extract_bit_exact(dsr):
    DEMSuperResolution, DSRConfig = dsr
    img, dem = synthetic_raster(300, 420, 7, hole=(100, 140, 200, 260))
    d = DEMSuperResolution(DSRConfig(image_size=64, stride=16, batch_size=4, tile_size=128))
    d.setImages(img, dem)
    d.padInputs()
    img_p, dem_p = tiler_ref.pad_inputs(img, dem, 64, 16, NOVAL)
    assert np.array_equal(d.img_padded.cpu().numpy(), img_p) and np.array_equal(d.dem_padded.cpu().numpy(), dem_p)
    assert d.generateTileList() == tiler_ref.tile_list(dem.shape, 128)
    org = d.patchOrigins(128, 0)
    assert [tuple(o) for o in org] == tiler_ref.patch_origins(128, 0, 128, 64, 16)
    # run the kernels directly on every origin of the tile
    import ctypes as C
    from moonsuperresolution_amd import _lib
    n = len(org)
    ox = torch.from_numpy(np.ascontiguousarray(org[:, 0])).cuda()
    oy = torch.from_numpy(np.ascontiguousarray(org[:, 1])).cuda()
    valid = torch.empty(n, dtype=torch.uint8, device="cuda")
    mm = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    rows, cols = d.dem_padded_shape
    rc = d._lib.msr_patch_stats(d._h, d.img_padded.data_ptr(), d.dem_padded.data_ptr(), rows, cols, ox.data_ptr(),
                                oy.data_ptr(), n, NOVAL, valid.data_ptr(), mm.data_ptr(), None)
    assert rc == 0
    out = torch.empty((n, 64, 64, 2), dtype=torch.float32, device="cuda")
    rc = d._lib.msr_extract_patches(d._h, d.img_padded.data_ptr(), d.dem_padded.data_ptr(), rows, cols,
                                    ox.data_ptr(), oy.data_ptr(), mm.data_ptr(), n, out.data_ptr(), None)
    assert rc == 0
    valid, mm, out = valid.cpu().numpy(), mm.cpu().numpy(), out.cpu().numpy()
    nvalid = 0
    for i, (xx, yy) in enumerate(org):
        ok, ip, dp = tiler_ref.get_patch(img_p, dem_p, int(xx), int(yy), 64, NOVAL)
        assert bool(valid[i]) == ok
        if ok:
            nvalid += 1
            patch, (lo, hi) = tiler_ref.normalize(ip, dp)
            assert mm[i, 2] == lo and mm[i, 3] == hi and mm[i, 0] == ip.min() and mm[i, 1] == ip.max()
            assert np.array_equal(out[i], patch)                   # bit-exact float32 normalisation
    assert 0 < nvalid < n
    d.close()


def test_nan_pixel_and_flat_patch_normalise_like_numpy(dsr):
    """process_full_tiles.py:307-309 is not guarded: a NaN pixel makes img_patch.min() / .max() NaN (NumPy
    propagates NaN, fminf would not) so the whole patch normalises to NaN, and a flat patch (max == min) divides
    0 by 0.  Both stay *valid* (NaN <= no_value is False).  The kernels do exactly the same, bit for bit."""
    DEMSuperResolution, DSRConfig = dsr
    img, dem = synthetic_raster(64, 192, 21)
    img[10, 70] = np.nan                 # patch 1 (x in [64,128)): NaN in the ortho only
    dem[:, 128:192] = -1234.5            # patch 2: flat DEM
    d = DEMSuperResolution(DSRConfig(image_size=64, stride=64, batch_size=4, tile_size=128))
    ox = torch.tensor([0, 64, 128], dtype=torch.int32, device="cuda")
    oy = torch.zeros(3, dtype=torch.int32, device="cuda")
    ti, td = torch.from_numpy(img).cuda(), torch.from_numpy(dem).cuda()
    valid = torch.empty(3, dtype=torch.uint8, device="cuda")
    mm = torch.empty((3, 4), dtype=torch.float32, device="cuda")
    assert d._lib.msr_patch_stats(d._h, ti.data_ptr(), td.data_ptr(), 64, 192, ox.data_ptr(), oy.data_ptr(), 3, NOVAL,
                                  valid.data_ptr(), mm.data_ptr(), None) == 0
    out = torch.empty((3, 64, 64, 2), dtype=torch.float32, device="cuda")
    assert d._lib.msr_extract_patches(d._h, ti.data_ptr(), td.data_ptr(), 64, 192, ox.data_ptr(), oy.data_ptr(),
                                      mm.data_ptr(), 3, out.data_ptr(), None) == 0
    out, mm = out.cpu().numpy(), mm.cpu().numpy()
    assert valid.cpu().numpy().tolist() == [1, 1, 1]
    with np.errstate(all="ignore"):
        for i in range(3):
            ok, ip, dp = tiler_ref.get_patch(img, dem, 64 * i, 0, 64, NOVAL)
            patch, (lo, hi) = tiler_ref.normalize(ip, dp)
            assert ok
            assert np.array_equal(out[i], patch.astype(np.float32), equal_nan=True)
            assert np.array_equal(mm[i], np.array([ip.min(), ip.max(), lo, hi], np.float32), equal_nan=True)
    assert np.isnan(out[1, :, :, 0]).all() and np.isfinite(out[1, :, :, 1]).all()     # NaN ortho channel, DEM intact
    assert np.isnan(out[2, :, :, 1]).all() and np.isfinite(out[0]).all()
    d.close()


def test_identity_model_map_bit_exact(dsr):
    """The reference's own known-answer check (process_full_tiles.py:139-143) end to end on the GPU tiler,
    against the NumPy oracle driven by the same (float32) identity model: every output bit-identical."""
    DEMSuperResolution, DSRConfig = dsr
    img, dem = synthetic_raster(200, 330, 3, hole=(90, 110, 140, 170))
    cfg = DSRConfig(image_size=64, stride=16, batch_size=4, tile_size=128)
    d = DEMSuperResolution(cfg, model=f32_identity)
    mean, std, good = d.processMap(img, dem)
    rm, rs, rg = tiler_ref.process_map(img, dem, f32_identity, 64, 16, 4, 128, NOVAL)
    assert np.array_equal(good, rg) and np.array_equal(mean, rm) and np.array_equal(std, rs)
    ok = good == 1
    assert np.abs(mean[ok] - dem[ok]).max() < 2e-3 * (dem[ok].max() - dem[ok].min()) and std[ok].max() < 0.05
    # the reference's literal identity lambda returns float64 for the zero-padded last batch: last-bit effects only
    rm64, _, rg64 = tiler_ref.process_map(img, dem, tiler_ref.identity_model, 64, 16, 4, 128, NOVAL)
    assert np.array_equal(rg64, good) and rel_linf(mean[ok], rm64[ok]) < 1e-6
    d.close()


def test_batch_composition_matches_reference(dsr):
    DEMSuperResolution, DSRConfig = dsr
    img, dem = synthetic_raster(100, 100, 4, hole=(0, 30, 0, 30))
    d = DEMSuperResolution(DSRConfig(image_size=64, stride=32, batch_size=4, tile_size=128), model=f32_identity)
    d.setImages(img, dem)
    d.padInputs()
    d.processTile(0, 0)
    img_p, dem_p = tiler_ref.pad_inputs(img, dem, 64, 32, NOVAL)
    _, calls = tiler_ref.process_tile(img_p, dem_p, 0, 0, f32_identity, 64, 32, 4, 128, NOVAL, return_batches=True)
    assert d.last_calls == calls
    d.close()


def test_generator_through_tiler_vs_oracle(dsr):
    """process_full_tiles over the HIP generator vs the oracle tiler over the oracle generator, same weights,
    same batch composition (gaugan_no_kl: no sampler noise)."""
    from moonsuperresolution_amd import Generator, make_weights
    from oracle import generator_ref
    DEMSuperResolution, DSRConfig = dsr
    w = make_weights("gaugan_no_kl", 64, seed=1234, bias_scale=0.05)
    img, dem = synthetic_raster(120, 120, 9)
    cfg = DSRConfig(image_size=64, stride=32, batch_size=4, tile_size=128)
    gen = Generator(64, 4, variant="gaugan_no_kl", weights=w)
    d = DEMSuperResolution(cfg, model=gen)
    mean, std, good = d.processMap(img, dem)
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    rm, rs, rg = tiler_ref.process_map(
        img, dem, lambda x, training=False: generator_ref.spade_call(x, wt, "gaugan_no_kl", dtype=torch.float32),
        64, 32, 4, 128, NOVAL)
    assert np.array_equal(good, rg) and good.any()
    ok = good == 1
    span = float(dem.max() - dem.min())
    assert np.abs(mean[ok] - rm[ok]).max() <= 1e-3 * span
    assert np.abs(std[ok] - rs[ok]).max() <= 1e-3 * span
    # the tile loop alternates its calls over two generator handles (pipeline=2, the default); one handle on one
    # stream must give exactly the same rasters (the kernels are deterministic, the clone has the same weights)
    d1 = DEMSuperResolution(cfg, model=gen, pipeline=1)
    m1, s1, g1 = d1.processMap(img, dem)
    assert np.array_equal(m1, mean) and np.array_equal(s1, std) and np.array_equal(g1, good)
    assert len(d._gens) == 2 and len(d1._gens) == 1
    d1.close()
    d.close()
    gen.close()


def test_sharded_map_with_the_real_process_tile(dsr):
    """distributed.process_map_sharded driven by DEMSuperResolution.processTile (device tensors, HIP generator) for
    simulated (rank, world) = (0..2, 3) without gather: the three partial maps add up to the single-process map bit
    for bit — tiles are independent units (process_full_tiles.py:313-325, 431-479)."""
    from moonsuperresolution_amd import Generator, make_weights
    from moonsuperresolution_amd.distributed import process_map_sharded, shard_tile_rows
    DEMSuperResolution, DSRConfig = dsr
    T = 128
    img, dem = synthetic_raster(300, 200, 17, hole=(150, 170, 20, 60))       # 3 tile rows x 2 tile columns
    gen = Generator(64, 4, variant="gaugan_no_kl", weights=make_weights("gaugan_no_kl", 64, seed=3, bias_scale=0.05))
    d = DEMSuperResolution(DSRConfig(image_size=64, stride=32, batch_size=4, tile_size=T), model=gen)
    single = d.processMap(img, dem)
    tiles = d.generateTileList()
    assert [len(shard_tile_rows(tiles, r, 3)) for r in range(3)] == [2, 2, 2]
    dev = torch.device("cuda", 0)
    one = process_map_sharded(dem.shape, T, tiles, d.processTile, 0, 1, device=dev)
    acc = [np.zeros_like(a) for a in single]
    for r in range(3):
        part = process_map_sharded(dem.shape, T, tiles, d.processTile, r, 3, gather=False, device=dev)
        for a, p in zip(acc, part):
            a += p
    for a, b, c in zip(single, one, acc):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    assert single[2].any()
    d.close(); gen.close()


def test_config3_geometry_at_size_and_one_tile_of_a_shard(dsr):
    """BASELINE configs[3]: the 15000 x 70000 raster of README.md:13 at the production settings of run_GAN.sh:24-26
    (S = 512, s = 64, T = 1024).  Canvas 16256 x 71552, 15 x 69 = 1035 tiles, tile rows dealt 8 / 7 over 2 ranks,
    4,4,4,3 over 4 and 2,2,2,2,2,2,2,1 over 8 (distributed.shard_tile_rows).  One tile of rank 3 of 8 is processed on the
    full-size canvas (patches cut at offsets beyond 2^31 bytes) and must equal, bit for bit, the same tile processed
    from a small raster that holds just its window: tiles are independent units (process_full_tiles.py:431-479)."""
    from moonsuperresolution_amd import Generator
    from moonsuperresolution_amd.distributed import shard_tile_rows
    DEMSuperResolution, DSRConfig = dsr
    S, s, T, B = 512, 64, 1024, 8
    rows, cols = 15000, 70000
    gen = Generator(S, B, variant="gaugan_no_kl", weights=1234)
    cfg = DSRConfig(image_size=S, stride=s, batch_size=B, tile_size=T)
    big = DEMSuperResolution(cfg, model=gen, pipeline=1)
    # the padded canvases, built on the device (a smooth field + noise; nodata outside the raster as padInputs leaves it)
    big.dem_shape = big.img_shape = (rows, cols)
    hp, wp = ((rows // 1024) + 1) * 1024 + 2 * (S - s), ((cols // 1024) + 1) * 1024 + 2 * (S - s)
    assert (hp, wp) == (16256, 71552)
    g = torch.Generator(device="cuda").manual_seed(5)
    yy = torch.arange(hp, device="cuda", dtype=torch.float32)[:, None]
    xx = torch.arange(wp, device="cuda", dtype=torch.float32)[None, :]
    big.dem_padded = -2000 + 600 * torch.sin(xx / 370.0) * torch.cos(yy / 530.0) + torch.randn((hp, wp), generator=g, device="cuda")
    big.img_padded = 0.5 + 0.3 * torch.cos(xx / 230.0) * torch.sin(yy / 310.0) + 0.05 * torch.randn((hp, wp), generator=g, device="cuda")
    halo = S - s
    for t in (big.dem_padded, big.img_padded):
        t[:halo] = NOVAL; t[:, :halo] = NOVAL; t[halo + rows:] = NOVAL; t[:, halo + cols:] = NOVAL
    big.dem_padded_shape = big.img_padded_shape = (hp, wp)
    tiles = big.generateTileList()
    assert len(tiles) == 1035 and len({y for _, y in tiles}) == 15 and len({x for x, _ in tiles}) == 69
    for world, want in ((2, [8, 7]), (4, [4, 4, 4, 3]), (8, [2, 2, 2, 2, 2, 2, 2, 1])):
        assert [len(shard_tile_rows(tiles, r, world)) // 69 for r in range(world)] == want
    mine = shard_tile_rows(tiles, 3, 8)
    assert len(mine) == 138 and mine[0] == (0, 6144)
    px, py = mine[68]                                    # last tile of rank 3's first row: x = 69632, byte offsets > 2^31
    assert (px, py) == (69632, 6144)
    mean, std, good = big.processTile(px, py)
    nv, ncall = big.last_counts
    assert 0 < nv < 529 and ncall == -(-nv // B)         # the raster ends inside this tile: part of its patches is nodata
    # the same tile from a raster that holds only its window
    win = T + 2 * halo
    small = DEMSuperResolution(cfg, model=gen, pipeline=1)
    small.dem_shape = small.img_shape = (T, T)
    small.dem_padded = big.dem_padded[py:py + win, px:px + win].contiguous()
    small.img_padded = big.img_padded[py:py + win, px:px + win].contiguous()
    small.dem_padded_shape = small.img_padded_shape = (win, win)
    m2, s2, g2 = small.processTile(0, 0)
    assert small.last_counts == (nv, ncall)
    assert torch.equal(good, g2) and torch.equal(mean, m2) and torch.equal(std, s2) and bool(good.any())
    big.close(); small.close(); gen.close()
    del big, small
    torch.cuda.empty_cache()


def test_full_size_stitch_properties(dsr):
    """BASELINE geometry S=512, s=64, T=1024 (529 patches): constant predictions stitch to a constant."""
    DEMSuperResolution, DSRConfig = dsr
    S, s, T = 512, 64, 1024
    d = DEMSuperResolution(DSRConfig(image_size=S, stride=s, batch_size=8, tile_size=T))
    n_side = len(range(0, T + S - s, s))
    keys = np.array([(ix * s, iy * s) for iy in range(n_side) for ix in range(n_side)], np.int32)
    n = len(keys)
    assert n == 529
    pred = torch.full((n, S, S), 0.25, dtype=torch.float32, device="cuda")      # -> 0.75 after +0.5
    mm = torch.tensor([[-100.0, 100.0]] * n, dtype=torch.float32, device="cuda")
    mean, std, good = d.rebuildTile(pred, torch.from_numpy(keys).cuda(), mm)
    assert good.all() and torch.all(mean == 50.0) and float(std.max()) < 1e-4
    # dropping every patch of the first two patch rows leaves the top strip un-reconstructed
    keep = keys[:, 1] >= 2 * s
    mean2, std2, good2 = d.rebuildTile(pred[: int(keep.sum())], torch.from_numpy(keys[keep]).cuda(), mm[: int(keep.sum())])
    assert good2[100:].all() and torch.all(mean2[good2.bool()] == 50.0)
    d.close()


@pytest.mark.parametrize("S,stride,B,T,shape,hole", [
    (64, 32, 3, 128, (150, 70), None),                 # raster narrower than a tile, batch size that never fills
    (64, 8, 16, 64, (90, 200), (20, 60, 100, 130)),    # dense overlap (stride S/8), small tiles, nodata block
    (128, 32, 5, 256, (300, 300), (0, 300, 140, 150)), # S=128, a nodata stripe through the whole raster
    (64, 64, 4, 128, (130, 130), (10, 20, 10, 20)),    # stride == image_size: no overlap at all
])
def test_identity_map_bit_exact_many_geometries(dsr, S, stride, B, T, shape, hole):
    """Randomised geometry sweep of the whole tiler + stitcher against the NumPy oracle (float32 identity model):
    padded canvas, tile list, validity, batch composition, normalisation, stitching, assembly — bit for bit."""
    DEMSuperResolution, DSRConfig = dsr
    img, dem = synthetic_raster(shape[0], shape[1], seed=S + stride, hole=hole)
    d = DEMSuperResolution(DSRConfig(image_size=S, stride=stride, batch_size=B, tile_size=T), model=f32_identity)
    mean, std, good = d.processMap(img, dem)
    rm, rs, rg = tiler_ref.process_map(img, dem, f32_identity, S, stride, B, T, NOVAL)
    assert mean.shape == shape and np.array_equal(good, rg)
    assert np.array_equal(mean, rm) and np.array_equal(std, rs)
    d.close()


def test_all_nodata_raster_gives_no_value_everywhere(dsr):
    DEMSuperResolution, DSRConfig = dsr
    img = np.full((100, 100), 0.5, np.float32)
    dem = np.full((100, 100), NOVAL, np.float32)
    d = DEMSuperResolution(DSRConfig(image_size=64, stride=32, batch_size=4, tile_size=128), model=f32_identity)
    mean, std, good = d.processMap(img, dem)
    assert not good.any() and (mean == NOVAL).all() and (std == NOVAL).all()
    d.close()


def test_files_in_files_out(dsr, tmp_path):
    """loadImages -> tiles -> rebuildMap -> saveGTiff on real files (process_full_tiles.py:568-587 without GDAL):
    the three products are LZW/predictor-2 GeoTIFFs carrying the input DEM's georeferencing and nodata, readable
    by libtiff, and hold exactly what the in-memory path computes."""
    import struct
    from PIL import Image
    from moonsuperresolution_amd import geotiff
    DEMSuperResolution, DSRConfig = dsr
    img, dem = synthetic_raster(180, 260, 12, hole=(60, 90, 100, 150))
    meta = {"geo": {33550: (12, 3, struct.pack("<3d", 5.0, 5.0, 0.0)),
                    33922: (12, 6, struct.pack("<6d", 0, 0, 0, 7000.0, 900.0, 0))}, "byteorder": "<"}
    src = tmp_path / "src"
    src.mkdir()
    geotiff.write_geotiff(str(src / "run-DRG.tif"), img, meta, nodata=NOVAL)
    geotiff.write_geotiff(str(src / "run-DEM.tif"), dem, meta, nodata=NOVAL)
    cfg = DSRConfig(image_size=64, stride=16, batch_size=4, tile_size=128, map_name="apollo", save_path=str(tmp_path / "out"),
                    source_folder_path=str(src))
    d = DEMSuperResolution(cfg, model=f32_identity)
    d.processFiles(preprocess=False)      # the DEM is used as read (preprocess has its own tests)
    rm, rs, rg = tiler_ref.process_map(img, dem, f32_identity, 64, 16, 4, 128, NOVAL)
    for name, ref in (("mean", rm), ("std", rs), ("good", rg)):
        arr, m = geotiff.read_geotiff(str(tmp_path / "out" / f"apollo_{name}.tiff"))
        assert np.array_equal(arr, ref.astype(np.float32)) and m["nodata"] == NOVAL
        assert geotiff.geotransform(m) == (7000.0, 5.0, 0.0, 900.0, 0.0, -5.0)
        assert m["dtype"] == (np.uint16 if name == "good" else np.float32)          # good is stored as UInt16 (:499-501)
        with Image.open(str(tmp_path / "out" / f"apollo_{name}.tiff")) as im:
            assert np.array_equal(np.array(im).astype(np.float32), arr)
    with pytest.raises(ValueError):
        d.saveGTiff(np.zeros((4, 4), np.float64), np.float64, "bad")
    bad = DEMSuperResolution(DSRConfig(image_size=64, stride=16, batch_size=4, tile_size=128, map_name="m",
                                       save_path=str(tmp_path), source_folder_path=str(tmp_path / "nowhere")), model=f32_identity)
    with pytest.raises(ValueError):
        bad.loadImages()
    d.close(); bad.close()
