"""CPU: host-side logic — weights, workload constants, the C ABI surface, sharding — no GPU compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from moonsuperresolution_amd import _lib, make_weights, weight_shapes, workload
from moonsuperresolution_amd.distributed import shard_tile_rows, tile_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_param_counts_match_baseline():
    assert workload.param_count("gaugan", 256) == 121_552_257       # 121.55 M
    assert workload.param_count("gaugan", 512) == 184_515_969       # 184.52 M
    assert abs(workload.param_count("pix2pix", 256) / 1e6 - 54.4) < 0.05


def test_flops_match_baseline():
    f512 = workload.spade_flops_per_patch(512)
    assert abs(f512["total"] / 1e9 - 702.607) < 1e-3
    assert abs(workload.spade_flops_per_patch(256)["total"] / 1e9 - 175.652) < 1e-3
    assert abs(workload.pix2pix_flops_per_patch() / 1e9 - 11.929) < 1e-3
    assert abs(f512["spade_gamma_beta"] / f512["total"] - 0.499) < 1e-3 and abs(f512["resblock"] / f512["total"] - 0.485) < 1e-3


def test_raster_geometry_matches_baseline():
    g = workload.raster_geometry((15000, 70000), 512, 64)
    assert (g["canvas_rows"], g["canvas_cols"], g["tiles"], g["patches_per_tile"], g["patches"]) == (16256, 71552, 1035, 529, 547515)
    assert workload.raster_geometry((15000, 70000), 256, 32)["patches"] == 1574235


def test_weights_are_deterministic_and_shaped():
    a = make_weights("gaugan", 64, seed=1234)
    b = make_weights("gaugan", 64, seed=1234)
    assert list(a) == list(weight_shapes("gaugan", 64))
    assert all(np.array_equal(a[k], b[k]) and a[k].dtype == np.float32 for k in a)
    assert a["gen.rb4.conv_3.kernel"].shape == (3, 3, 1024, 512) and "gen.rb1.conv_3.kernel" not in a
    assert a["gen.head.kernel"].shape == (4, 4, 128, 1) and (a["gen.head.bias"] == 0).all()
    lim = np.sqrt(6.0 / (9 * 128 + 9 * 1024))
    assert np.abs(a["gen.rb1.spade_1.conv_gamma.kernel"]).max() <= lim
    c = make_weights("gaugan", 64, seed=1)
    assert not np.array_equal(a["gen.dense.kernel"], c["gen.dense.kernel"])
    with pytest.raises(ValueError):
        weight_shapes("gaugan", 100)


def test_library_exports_every_declared_symbol(hip_lib):
    header = open(os.path.join(ROOT, "include", "moonsr.h")).read()
    declared = set(re.findall(r"\b(msr_[a-z_0-9]+)\s*\(", header))
    declared -= {"msr_handle", "msr_config", "msr_status", "msr_variant", "msr_kernel_stat"}
    assert len(declared) >= 18
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(hip_lib, name), name
    assert hip_lib.msr_abi_version() == 1


def test_create_fails_loudly_without_gpu(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cfg = _lib.MsrConfig(64, 2, 256, 0, 0, 0)
    h = C.c_void_p()
    rc = hip_lib.msr_create(C.byref(cfg), C.byref(h))
    assert rc == _lib.MSR_ERR_DEVICE and not h.value
    assert b"no CPU fallback" in hip_lib.msr_last_error(None) or b"HIP" in hip_lib.msr_last_error(None)
    from moonsuperresolution_amd import Generator
    with pytest.raises(RuntimeError):
        Generator(64, 2, weights={})


def test_create_rejects_bad_config(hip_lib):
    h = C.c_void_p()
    for cfg in (_lib.MsrConfig(100, 2, 256, 0, 0, 0), _lib.MsrConfig(64, 0, 256, 0, 0, 0),
                _lib.MsrConfig(64, 2, 256, 9, 0, 0), _lib.MsrConfig(128, 1, 256, 3, 0, 0)):
        assert hip_lib.msr_create(C.byref(cfg), C.byref(h)) == _lib.MSR_ERR_INVALID


def test_tile_row_sharding():
    tiles = [(xx, yy) for yy in range(0, 15000, 1024) for xx in range(0, 70000, 1024)]
    assert len(tile_rows(tiles)) == 15
    sizes = []
    seen = []
    for r in range(8):
        mine = shard_tile_rows(tiles, r, 8)
        sizes.append(len(tile_rows(mine)))
        seen += mine
    assert sizes == [2, 2, 2, 2, 2, 2, 2, 1] and sorted(seen) == sorted(tiles)
    assert shard_tile_rows(tiles, 0, 1) == tiles
    assert shard_tile_rows(tiles[:69], 1, 2) == []      # fewer rows than ranks: some ranks idle


def test_e4m3_quantiser_matches_torch():
    """msr_quantize_e4m3 (what msr_load_weight applies in the fp8 mode) against torch.float8_e4m3fn on a sweep that
    covers every binade, the subnormals, the rounding ties, the saturation at 448 and both signs."""
    import ctypes as C
    import torch
    from moonsuperresolution_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    vals = np.concatenate([
        rng.standard_normal(20000) * np.exp(rng.uniform(-12, 7, 20000)),
        np.array([0.0, -0.0, 448.0, 449.0, 1e9, -1e9, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10, 2.0 ** -6, 464.0, 479.9, 17.0, 18.0, 19.0]),
        (np.arange(0, 256) + 0.5) * 2.0 ** -9, np.arange(1, 32) * 0.0625 + 0.03125]).astype(np.float32)
    out = np.empty(vals.size, np.uint8)
    assert lib.msr_quantize_e4m3(vals.ctypes.data_as(C.c_void_p), vals.size, out.ctypes.data_as(C.c_void_p)) == vals.size
    ref = torch.from_numpy(np.clip(vals, -448, 448)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got_f = torch.from_numpy(out).view(torch.float8_e4m3fn).float().numpy()
    ref_f = torch.from_numpy(ref).view(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(got_f, ref_f)              # compare values: +0 and -0 are the same weight


def test_f16c6_image_helpers_round_trip_on_cpu():
    """The host restatement of the f16c6 chunk image (csrc/kernels.h PREC_F16C6: fp16 main piece + two e2m3 pieces with a
    power-of-two scale per 32-channel block): decode(pack(x)) returns the pieces bit for bit, the block scale is the
    smallest power of two >= max|x| / 7.5, h6 carries e2m3's 3 mantissa bits on that scale and hi + l6 recovers x to ~2^-15."""
    import torch
    from moonsuperresolution_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 3, 5, 96), generator=g) * torch.logspace(-3, 1, 96)
    x[0, 0, 0, :32] = 0.0                                   # an all-zero block: scale 2^0, codes 0
    img, (hi, h6, l6) = ops.f16c6_activation_image(x)
    assert img.shape == x.shape and img.dtype == torch.float32
    d_hi, d_h6, d_l6 = ops.f16c6_decode(img)
    assert torch.equal(d_hi, hi) and torch.equal(d_h6, h6) and torch.equal(d_l6, l6)
    raw = img.contiguous().view(torch.uint8).reshape(-1, 128)
    amax = x.double().abs().reshape(-1, 32).amax(-1)
    E = raw[:, 88].double() - 127
    nz = amax > 0
    assert bool((2.0 ** E[nz] >= amax[nz] / 7.5).all()) and bool((2.0 ** (E[nz] - 1) < amax[nz] / 7.5).all())
    assert bool((raw[~nz][:, 88] == 127).all()) and bool((raw[:, 120].int() == raw[:, 88].int() - 11).all())
    assert int(raw[:, 89:96].max()) == 0 and int(raw[:, 121:128].max()) == 0
    blk = amax.reshape(x.shape[:-1] + (3, 1)).expand(x.shape[:-1] + (3, 32)).reshape(x.shape).clamp_min(1e-30)
    assert float(((h6 - x.double()).abs() / blk).max()) <= 1 / 15 + 1e-9            # half an e2m3 step of the top binade
    assert float(((hi + l6 - x.double()).abs() / blk).max()) <= 2.0 ** -14
    w = torch.randn((9, 64, 64), generator=g) * 0.05
    wimg, (wh, w6, wl) = ops.f16c6_weight_image(w)
    wraw = wimg.contiguous().view(torch.uint8).reshape(9, 64, 2, 128)
    assert bool((wraw[..., 88] == wraw[0, :, 0, 88][None, :, None]).all())           # one scale per output channel and piece
    assert float((wh + wl - w.double()).abs().max()) <= 2.0 ** -14 * float(w.abs().max())
