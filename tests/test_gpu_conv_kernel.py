"""GPU (-m gpu): kernel-level parity of conv_igemm_f32 (all epilogues, both tiles, both strides) against a
plain PyTorch fp32/fp64 reference of the same op.  Tolerance: rel L-inf <= 1e-5 vs fp64 (fp32 accumulation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_linf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(hip_lib):
    assert torch.cuda.is_available()
    from moonsuperresolution_amd import ops
    c = ops.OpContext()
    yield c
    c.close()


def ref_conv(x, w, b, stride):
    """TF SAME conv on NHWC in float64 on the CPU."""
    x, w, b = x.double().cpu(), w.double().cpu(), b.double().cpu()
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (1, 1, 1, 1)) if stride == 1 else F.pad(xn, (0, 1, 0, 1))
    return F.conv2d(xn, w.permute(3, 2, 0, 1), b, stride=stride).permute(0, 2, 3, 1)


@pytest.mark.parametrize("B,r,cin,cout,stride,tile", [
    (2, 16, 64, 128, 1, 0), (2, 16, 64, 128, 1, 1), (3, 8, 32, 64, 1, 1), (1, 4, 64, 64, 1, 1),
    (5, 2, 32, 128, 1, 0), (2, 16, 64, 128, 2, 1), (2, 32, 32, 128, 2, 0), (16, 1, 64, 64, 1, 1),
    (2, 8, 128, 128, 1, 1 + 256 * 4), (1, 4, 256, 128, 1, 0 + 256 * 8), (2, 16, 64, 64, 2, 1 + 256 * 3)])
def test_conv_bias(ctx, B, r, cin, cout, stride, tile):
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + r)
    rin = r * stride
    x = torch.randn((B, rin, rin, cin), generator=g).cuda()
    w = (torch.randn((3, 3, cin, cout), generator=g) / np.sqrt(9 * cin)).cuda()
    b = torch.randn(cout, generator=g).cuda()
    y = ops.conv3x3(ctx, ops.pad_nhwc(x), ops.kernel_layout(w), b, r, stride=stride, tile=tile)
    assert rel_linf(y.cpu().numpy(), ref_conv(x, w, b, stride).numpy()) <= 1e-5


def test_conv_residual_bf16x3_pingpong_ksplit(ctx):
    """Residual epilogue through the K-split ping-pong launch (rb2 conv_2 at S = 512, B = 8: 64 tiles x 4 K ranges)."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    B, r, cin, cout = 2, 16, 256, 128
    x = torch.randn((B, r, r, cin), generator=g).cuda()
    w = (torch.randn((3, 3, cin, cout), generator=g) / 48).cuda()
    b = torch.randn(cout, generator=g).cuda()
    res = torch.randn((B, r >> 1, r >> 1, cout), generator=g).cuda()
    y = ops.conv3x3(ctx, ops.split_bf16(ctx, ops.pad_nhwc(x)), ops.split_bf16(ctx, ops.kernel_layout(w)), b, r,
                    epilogue=ops.EPI_RES, aux=res, aux_shift=1, tile=5 + 256 * 4, precision="bf16x3")
    up = res.double().cpu().repeat_interleave(2, 1).repeat_interleave(2, 2)
    assert rel_linf(y.cpu().numpy(), (ref_conv(x, w, b, 1) + up).numpy()) <= 5e-5


@pytest.mark.parametrize("shift,tile", [(0, 0), (1, 0), (1, 1), (1, 1 + 256 * 2)])
def test_conv_residual_with_upsample_fold(ctx, shift, tile):
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(7)
    B, r, cin, cout = 2, 16, 64, 128
    x = torch.randn((B, r, r, cin), generator=g).cuda()
    w = (torch.randn((3, 3, cin, cout), generator=g) / 24).cuda()
    b = torch.randn(cout, generator=g).cuda()
    res = torch.randn((B, r >> shift, r >> shift, cout), generator=g).cuda()
    y = ops.conv3x3(ctx, ops.pad_nhwc(x), ops.kernel_layout(w), b, r, epilogue=ops.EPI_RES, aux=res, aux_shift=shift,
                    tile=tile)
    up = res.double().cpu()
    if shift:
        up = up.repeat_interleave(2, 1).repeat_interleave(2, 2)
    assert rel_linf(y.cpu().numpy(), (ref_conv(x, w, b, 1) + up).numpy()) <= 1e-5


@pytest.mark.parametrize("shift,tile,C", [(0, 0, 64), (1, 0, 128), (0, 1, 32), (1, 1 + 256 * 4, 64)])
def test_conv_spade_epilogue(ctx, shift, tile, C):
    """gamma/beta as one N=2C GEMM + leaky_relu(gamma * (x-mean)/std + beta) into a zero-bordered tensor."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(11)
    B, r = 2, 16
    h = torch.relu(torch.randn((B, r, r, 128), generator=g)).cuda()
    wg = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    wb = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    bg, bb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    x = (3 + 2 * torch.randn((B, r >> shift, r >> shift, C), generator=g)).cuda()
    mean = x.mean((0, 1, 2)).contiguous()
    std = torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-5).contiguous()
    w, bias = ops.spade_layout(wg, wb, bg, bb)
    y = ops.conv3x3(ctx, ops.pad_nhwc(h), w, bias, r, epilogue=ops.EPI_SPADE, aux=x, aux_shift=shift, mean=mean,
                    std=std, out_padded=True, tile=tile)
    xr = x.double().cpu()
    if shift:
        xr = xr.repeat_interleave(2, 1).repeat_interleave(2, 2)
    v = ref_conv(h, wg, bg, 1) * ((xr - mean.double().cpu()) / std.double().cpu()) + ref_conv(h, wb, bb, 1)
    v = torch.where(v >= 0, v, 0.2 * v)
    yc = y.cpu()
    assert rel_linf(yc[:, 1:-1, 1:-1].numpy(), v.numpy()) <= 1e-5
    assert float(yc[:, 0].abs().max()) == 0 and float(yc[:, :, -1].abs().max()) == 0     # the border stays zero


def test_untileable_shape_is_rejected(ctx):
    from moonsuperresolution_amd import ops
    x = torch.zeros((1, 14, 14, 32), device="cuda")       # rout = 12 is not a power of two
    w = torch.zeros((9, 64, 32), device="cuda")
    with pytest.raises(ValueError):
        ops.conv3x3(ctx, x, w, torch.zeros(64, device="cuda"), 12)


@pytest.mark.parametrize("frag", [0, 0x40])
@pytest.mark.parametrize("B,r,cin,cout,stride,tile", [
    (2, 16, 64, 128, 1, 0), (2, 16, 64, 128, 1, 1), (3, 8, 32, 64, 1, 1), (2, 16, 64, 128, 2, 1),
    (2, 8, 128, 128, 1, 1 + 256 * 4), (16, 1, 64, 64, 1, 1), (2, 16, 64, 128, 1, 3), (1, 32, 128, 256, 1, 3),
    (2, 16, 64, 128, 1, 4), (1, 32, 128, 256, 1, 4), (3, 16, 128, 256, 1, 4),
    (2, 16, 64, 128, 1, 5), (1, 32, 128, 256, 1, 5), (3, 16, 128, 256, 1, 5),
    (3, 128, 64, 256, 1, 5), (16, 64, 64, 256, 1, 5),    # 384 / 512 tiles on 256 persistent workgroups: 1-2 tiles each
    (2, 16, 128, 256, 1, 5 + 256 * 2), (1, 32, 256, 128, 1, 5 + 256 * 4), (3, 16, 1024, 128, 1, 5 + 256 * 16),
    (8, 16, 512, 1024, 1, 5 + 256 * 4)])                 # K-split ping-pong: (K range, tile) work items + split-K pass
def test_conv_bf16x3(ctx, B, r, cin, cout, stride, tile, frag):
    """3-term split-bf16 products, fp32 accumulation: error bound ~3*2^-18 per product -> rel L-inf <= 5e-5.
    Three kernels: LDS-staged weights (tiles 0/1), weights in VGPRs (| 0x40), LDS-staged input halo (tile 3 on the
    32x32x16 MFMA, tile 4 on the 16x16x32 MFMA, tile 5 = the 512-thread ping-pong form of tile 4)."""
    from moonsuperresolution_amd import ops
    if frag and (tile & 0x3F) in (3, 4, 5):
        pytest.skip("the halo kernel stages its weights through LDS")
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + r + 1)
    rin = r * stride
    x = torch.randn((B, rin, rin, cin), generator=g).cuda()
    w = (torch.randn((3, 3, cin, cout), generator=g) / np.sqrt(9 * cin)).cuda()
    b = torch.randn(cout, generator=g).cuda()
    xs = ops.split_bf16(ctx, ops.pad_nhwc(x))
    ws = ops.weights_bf16x3(ops.kernel_layout(w)) if frag else ops.split_bf16(ctx, ops.kernel_layout(w))
    y = ops.conv3x3(ctx, xs, ws, b, r, stride=stride, tile=tile | frag, precision="bf16x3")
    err = rel_linf(y.cpu().numpy(), ref_conv(x, w, b, stride).numpy())
    assert err <= 5e-5, err


def unsplit(t):
    """split-bf16 chunk image -> (hi + lo) float32 values, same shape (innermost dim % 32 == 0)."""
    u = t.contiguous().view(torch.int16).reshape(-1, 2, 32).to(torch.int32) & 0xFFFF
    hi = (u[:, 0] << 16).view(torch.float32)
    lo = (u[:, 1] << 16).view(torch.float32)
    return hi.reshape(t.shape), lo.reshape(t.shape)


def test_split_bf16_words(ctx):
    from moonsuperresolution_amd import ops
    x = torch.randn(4096, device="cuda") * torch.logspace(-6, 6, 4096, device="cuda")
    hi, lo = unsplit(ops.split_bf16(ctx, x))
    assert torch.equal(hi, x.to(torch.bfloat16).float())                       # hi = round-to-nearest-even bf16
    assert torch.equal(lo, (x - hi).to(torch.bfloat16).float())
    assert float(((hi + lo) - x).abs().max() / x.abs().max()) < 2 ** -16
    z = ops.split_bf16(ctx, torch.zeros(64, device="cuda"))
    assert float(z.abs().max()) == 0.0                                         # the zero border stays zero


@pytest.mark.parametrize("tile,B,r,C,shift", [
    (0, 2, 16, 64, 0), (3, 2, 16, 64, 0), (4, 2, 16, 64, 0), (5, 2, 16, 64, 0),
    (5, 3, 128, 128, 0),      # 384 ping-pong tiles on 256 persistent workgroups: 1-2 tiles each (the bench regime)
    (5, 16, 64, 128, 1),      # 512 tiles, x read through the folded 2x up-sample (rb5 spade_1 at S=256, B=16)
    (5 + 256 * 2, 4, 16, 256, 1),     # K-split ping-pong (2 ranges of one chunk pair each) + split-K SPADE pass
    (4, 3, 32, 256, 1)])      # 192 tiles of the 2-workgroups-per-CU halo form
def test_conv_spade_epilogue_bf16x3_split_output(ctx, tile, B, r, C, shift):
    """The SPADE epilogue of the bf16x3 kernels writing split-bf16 words into a zero-bordered tensor, including the
    persistent ping-pong kernel with more tiles than workgroups (spade.py:19-24 fused into the gamma|beta GEMM)."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(13 + B + r)
    h = torch.relu(torch.randn((B, r, r, 128), generator=g)).cuda()
    wg = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    wb = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    bg, bb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    x = (3 + 2 * torch.randn((B, r >> shift, r >> shift, C), generator=g)).cuda()
    mean = x.mean((0, 1, 2)).contiguous()
    std = torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-5).contiguous()
    w, bias = ops.spade_layout(wg, wb, bg, bb)
    y = ops.conv3x3(ctx, ops.split_bf16(ctx, ops.pad_nhwc(h)), ops.split_bf16(ctx, w), bias, r, epilogue=ops.EPI_SPADE,
                    aux=x, aux_shift=shift, mean=mean, std=std, out_padded=True, tile=tile, precision="bf16x3",
                    out_split=True)
    hi, lo = unsplit(y)
    val = hi + lo
    xr = x.double().cpu()
    if shift:
        xr = xr.repeat_interleave(2, 1).repeat_interleave(2, 2)
    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 1 else 16))
    v = ref_conv(h, wg, bg, 1) * ((xr - mean.double().cpu()) / std.double().cpu()) + ref_conv(h, wb, bb, 1)
    v = torch.where(v >= 0, v, 0.2 * v)
    assert rel_linf(val.cpu()[:, 1:-1, 1:-1].numpy(), v.numpy()) <= 5e-5
    assert float(val.cpu()[:, 0].abs().max()) == 0 and float(val.cpu()[:, :, -1].abs().max()) == 0


@pytest.mark.parametrize("B,r,C,shift", [(2, 16, 64, 0), (3, 128, 128, 0), (16, 64, 128, 1)])
def test_conv_spade_epilogue_f16x2(ctx, B, r, C, shift):
    """The opt-in 2-term form of the gamma|beta conv (MSR_FLAG_GB_F16X2; persistent ping-pong tile only): activation
    = two fp16 halves, weight = ONE fp16, products on v_mfma_f32_16x16x32_f16 with fp32 accumulation.  The weight's
    rounding (2^-12 relative per product, random sign over K = 1152 terms) bounds the result: rel L-inf <= 5e-4
    against the float64 reference (observed ~1e-4); against a reference computed with the fp16-rounded weights the
    kernel must agree like the 3-term one does (<= 5e-5): the activation split loses nothing."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(29 + B + r)
    h = torch.relu(torch.randn((B, r, r, 128), generator=g)).cuda()
    wg = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    wb = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    bg, bb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    x = (3 + 2 * torch.randn((B, r >> shift, r >> shift, C), generator=g)).cuda()
    mean = x.mean((0, 1, 2)).contiguous()
    std = torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-5).contiguous()
    w, bias = ops.spade_layout(wg, wb, bg, bb)
    y = ops.conv3x3(ctx, ops.split_f16(ops.pad_nhwc(h)), ops.split_f16(w), bias, r, epilogue=ops.EPI_SPADE, aux=x,
                    aux_shift=shift, mean=mean, std=std, out_padded=True, tile=ops.TILE_PP | ops.TILE_F16X2,
                    precision="bf16x3", out_split=True)
    hi, lo = unsplit(y)
    val = (hi + lo).cpu()[:, 1:-1, 1:-1].numpy()
    xr = x.double().cpu()
    if shift:
        xr = xr.repeat_interleave(2, 1).repeat_interleave(2, 2)
    xn = (xr - mean.double().cpu()) / std.double().cpu()

    def ref(wg_, wb_):
        v = ref_conv(h, wg_, bg, 1) * xn + ref_conv(h, wb_, bb, 1)
        return torch.where(v >= 0, v, 0.2 * v).numpy()

    e_full = rel_linf(val, ref(wg, wb))
    e_rounded = rel_linf(val, ref(wg.half().float(), wb.half().float()))
    print("f16x2 kernel: rel Linf vs fp64", e_full, " vs fp64 with fp16-rounded weights", e_rounded)
    assert e_full <= 5e-4, e_full
    assert e_rounded <= 5e-5, e_rounded


@pytest.mark.parametrize("B,r,cin,cout", [(2, 16, 256, 128), (1, 32, 512, 256), (3, 16, 128, 128), (20, 64, 256, 128),
                                          (1, 16, 128, 128), (2, 32, 128, 256), (17, 64, 128, 128)])
def test_conv_fp8_exact_on_quantised_operands(ctx, B, r, cin, cout):
    """The fp8 form of the ping-pong conv (declared non-parity mode): fp8 e4m3 weights with a power-of-two scale per
    output channel in the MFMA's e8m0 scale operand, bf8 e5m2 activations, K = 128 per instruction, fp32 accumulation.
    Operands are quantised HERE with torch's float8 types; against a float64 conv of the de-quantised operands the
    kernel must be exact up to the accumulation (<= 5e-5, the bound of the bf16x3 kernels): that pins the byte layout, the operand pairing of the
    128-deep MFMA, the one-chunk form (cin = 128: two tiles per unrolled body, odd and even tile counts) and the scale plumbing.  The quantisation error itself is what
    the mode declares (tests/test_gpu_baseline_configs.py)."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(41 + B + r)
    x = torch.randn((B, r, r, cin), generator=g).cuda()
    w = (torch.randn((3, 3, cin, cout), generator=g) / np.sqrt(9 * cin) * torch.logspace(-2, 1, cout)).cuda()
    b = torch.randn(cout, generator=g).cuda()
    xb, xdq = ops.bf8_activation_image(ops.pad_nhwc(x))
    wb, wexp, wdq = ops.fp8_weight_image(ops.kernel_layout(w))
    y = ops.conv3x3_fp8(ctx, xb, wb, wexp, b, r)
    w_hwio = wdq.permute(0, 2, 1).reshape(3, 3, cin, cout)
    ref = ref_conv(xdq[:, 1:-1, 1:-1], w_hwio, b, 1)
    err = rel_linf(y.cpu().numpy(), ref.numpy())
    print("fp8 conv vs fp64 conv of the de-quantised operands: rel Linf", err)
    assert err <= 5e-5, err
    # and the declared error of the quantisation itself, for the record
    print("fp8 conv vs unquantised fp64 conv: rel Linf", rel_linf(y.cpu().numpy(), ref_conv(x, w, b, 1).numpy()))


def test_conv_fp8_spade_epilogue_bf8_output(ctx):
    """SPADE epilogue of the fp8 gamma|beta conv writing bf8 bytes for an fp8 consumer, and
    split-bf16 words for a bf16x3 consumer."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(43)
    B, r, C, shift = 3, 32, 128, 1
    h = torch.relu(torch.randn((B, r, r, 128), generator=g)).cuda()
    wg = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    wb_ = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    bg, bb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    x = (3 + 2 * torch.randn((B, r >> shift, r >> shift, C), generator=g)).cuda()
    mean = x.mean((0, 1, 2)).contiguous()
    std = torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-5).contiguous()
    w, bias = ops.spade_layout(wg, wb_, bg, bb)
    hb, hdq = ops.bf8_activation_image(ops.pad_nhwc(h))
    wq, wexp, wdq = ops.fp8_weight_image(w)
    # reference on the de-quantised operands: un-interleave the (32 gamma | 32 beta) rows
    c = torch.arange(C)
    rows_g = ((c // 32) * 64 + (c % 32)).cuda()
    wdq_hwio = wdq.permute(0, 2, 1).reshape(3, 3, 128, 2 * C)
    xr = x.double().cpu().repeat_interleave(2, 1).repeat_interleave(2, 2)
    xn = (xr - mean.double().cpu()) / std.double().cpu()
    v = ref_conv(hdq[:, 1:-1, 1:-1], wdq_hwio[..., rows_g], bg, 1) * xn + ref_conv(hdq[:, 1:-1, 1:-1], wdq_hwio[..., rows_g + 32], bb, 1)
    v = torch.where(v >= 0, v, 0.2 * v)
    y32 = ops.conv3x3_fp8(ctx, hb, wq, wexp, bias, r, epilogue=ops.EPI_SPADE, aux=x, aux_shift=shift, mean=mean, std=std,
                          out_padded=True, out_mode=0)
    assert rel_linf(y32.cpu()[:, 1:-1, 1:-1].numpy(), v.numpy()) <= 5e-5
    ys = ops.conv3x3_fp8(ctx, hb, wq, wexp, bias, r, epilogue=ops.EPI_SPADE, aux=x, aux_shift=shift, mean=mean, std=std,
                         out_padded=True, out_mode=1)
    hi, lo = unsplit(ys)
    assert rel_linf((hi + lo).cpu()[:, 1:-1, 1:-1].numpy(), v.numpy()) <= 5e-5
    y8 = ops.conv3x3_fp8(ctx, hb, wq, wexp, bias, r, epilogue=ops.EPI_SPADE, aux=x, aux_shift=shift, mean=mean, std=std,
                         out_padded=True, out_mode=3)
    assert y8.shape == (B, r + 2, r + 2, 128)
    got = y8[..., :C].view(torch.float8_e5m2).float().cpu()[:, 1:-1, 1:-1]
    want = v.float().to(torch.float8_e5m2).float()
    mism = (got != want)
    assert float(mism.float().mean()) <= 2e-3                    # values on a rounding boundary may fall either way
    ulp = torch.maximum(want.abs() * 0.25, torch.tensor(2.0 ** -16))               # one bf8 step (2 mantissa bits; subnormal floor)
    assert bool(((got - want).abs() <= ulp * 1.001)[mism].all())
    assert int(y8[:, 0].max()) == 0 and int(y8[:, :, -1].max()) == 0     # the border stays zero


def _ref_f16c(xparts, wparts, bias, cin, cout):
    """What the f16c kernel computes, in float64: x_hi*w_hi + x_h8*w_lo8 + x_lo8*w_h8 (+ bias)."""
    (xh, x8, xl), (wh, w8, wl) = xparts, wparts
    hwio = lambda t: t.permute(0, 2, 1).reshape(3, 3, cin, cout)   # noqa: E731
    zero = torch.zeros(cout, dtype=torch.float64)
    return (ref_conv(xh, hwio(wh), bias, 1) + ref_conv(x8, hwio(wl), zero, 1) + ref_conv(xl, hwio(w8), zero, 1))


@pytest.mark.parametrize("B,r,cin,cout", [(2, 16, 64, 128), (1, 32, 256, 256), (3, 64, 128, 128), (9, 32, 128, 512)])
def test_conv_f16c(ctx, B, r, cin, cout):
    """fp16 main term + fp8 cross terms (MSR_FLAG_F16C): against the float64 evaluation of exactly those three terms on the
    de-quantised operands the kernel is exact up to accumulation (<= 5e-5: pins the chunk image, the pairing of the
    pieces of two taps in one 128-deep MFMA and the per-lane scales); against the float64 conv of the ORIGINAL operands it
    holds 2e-4 at kernel level (per-product error ~2^-15; weights spanning three decades)."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(51 + B + r)
    x = torch.randn((B, r, r, cin), generator=g).cuda()
    w = (torch.randn((3, 3, cin, cout), generator=g) / np.sqrt(9 * cin) * torch.logspace(-2, 1, cout)).cuda()
    b = torch.randn(cout, generator=g).cuda()
    ximg, xparts = ops.f16c_activation_image(ops.pad_nhwc(x))
    wimg, wexp, wparts = ops.f16c_weight_image(ops.kernel_layout(w))
    y = ops.conv3x3_f16c(ctx, ximg, wimg, wexp, b, r).cpu().numpy()
    xin = tuple(t[:, 1:-1, 1:-1].cpu() for t in xparts)
    emu = _ref_f16c(xin, tuple(t.cpu() for t in wparts), b, cin, cout).numpy()
    e_emu, e_true = rel_linf(y, emu), rel_linf(y, ref_conv(x, w, b, 1).numpy())
    print("f16c conv: vs its own three terms in fp64", e_emu, " vs the fp64 conv", e_true)
    assert e_emu <= 5e-5, e_emu
    assert e_true <= 2e-4, e_true


def test_conv_f16c_spade_epilogue_writes_the_chunk_image(ctx):
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(53)
    B, r, C, shift = 3, 32, 128, 1
    h = torch.relu(torch.randn((B, r, r, 128), generator=g)).cuda()
    wg = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    wb_ = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    bg, bb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    x = (3 + 2 * torch.randn((B, r >> shift, r >> shift, C), generator=g)).cuda()
    mean = x.mean((0, 1, 2)).contiguous()
    std = torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-5).contiguous()
    w, bias = ops.spade_layout(wg, wb_, bg, bb)
    himg, _ = ops.f16c_activation_image(ops.pad_nhwc(h))
    wimg, wexp, _ = ops.f16c_weight_image(w)
    xr = x.double().cpu().repeat_interleave(2, 1).repeat_interleave(2, 2)
    v = ref_conv(h, wg, bg, 1) * ((xr - mean.double().cpu()) / std.double().cpu()) + ref_conv(h, wb_, bb, 1)
    v = torch.where(v >= 0, v, 0.2 * v)
    y32 = ops.conv3x3_f16c(ctx, himg, wimg, wexp, bias, r, epilogue=ops.EPI_SPADE, aux=x, aux_shift=shift, mean=mean, std=std,
                           out_padded=True, out_mode=0)
    assert rel_linf(y32.cpu()[:, 1:-1, 1:-1].numpy(), v.numpy()) <= 2e-4
    yc = ops.conv3x3_f16c(ctx, himg, wimg, wexp, bias, r, epilogue=ops.EPI_SPADE, aux=x, aux_shift=shift, mean=mean, std=std,
                          out_padded=True, out_mode=4)
    hi, h8, lo8 = (t.cpu()[:, 1:-1, 1:-1] for t in ops.f16c_decode(yc))
    want = y32.cpu()[:, 1:-1, 1:-1].double()
    scale = float(want.abs().max())
    assert float((hi - want).abs().max()) <= 2.0 ** -11 * scale                 # hi is the fp16 rounding of the value
    assert float((hi + lo8 - want).abs().max()) <= 2.0 ** -14 * scale           # hi + lo8 recovers ~15 bits
    assert float(((h8 - want).abs() / want.abs().clamp_min(2.0 ** -6)).max()) <= 2.0 ** -4 * 1.01   # h8 = e4m3 of the value
    full = ops.f16c_decode(yc)
    assert float(full[0][:, 0].abs().max()) == 0 and float(full[1][:, :, -1].abs().max()) == 0     # the border stays zero


@pytest.mark.parametrize("B,r,cin,cout,res", [(1, 32, 256, 256, False), (3, 64, 128, 128, True), (9, 32, 128, 512, False)])
def test_conv_f16c6(ctx, B, r, cin, cout, res):
    """fp16 main term + fp6 e2m3 cross pieces with block scales (PREC_F16C6, the stream kernel): exact up to accumulation
    against the float64 evaluation of its own three terms on the de-quantised operands (pins the 6-bit packing, the 32-byte
    half a lane reads, the pairing of halves and taps in the 128-deep MFMA and the scales read from the images), and 2e-4
    against the float64 conv of the original operands."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(61 + B + r)
    x = (torch.randn((B, r, r, cin), generator=g) * torch.logspace(-1.5, 0.5, cin)).cuda()
    w = (torch.randn((3, 3, cin, cout), generator=g) / np.sqrt(9 * cin) * torch.logspace(-2, 1, cout)).cuda()
    b = torch.randn(cout, generator=g).cuda()
    skip = torch.randn((B, r, r, cout), generator=g).cuda() if res else None
    ximg, (xh, x6, xl) = ops.f16c6_activation_image(ops.pad_nhwc(x))
    wimg, (wh, w6, wl) = ops.f16c6_weight_image(ops.kernel_layout(w))
    y = ops.conv3x3_f16c(ctx, ximg, wimg, None, b, r, epilogue=ops.EPI_RES if res else ops.EPI_BIAS, aux=skip).cpu().numpy()
    hwio = lambda t: t.cpu().permute(0, 2, 1).reshape(3, 3, cin, cout)     # noqa: E731
    zero = torch.zeros(cout, dtype=torch.float64)
    cut = lambda t: t[:, 1:-1, 1:-1].cpu()                                 # noqa: E731
    emu = ref_conv(cut(xh), hwio(wh), b, 1) + ref_conv(cut(x6), hwio(wl), zero, 1) + ref_conv(cut(xl), hwio(w6), zero, 1)
    true = ref_conv(x, w, b, 1)
    if res:
        emu = emu + skip.double().cpu()
        true = true + skip.double().cpu()
    e_emu, e_true = rel_linf(y, emu.numpy()), rel_linf(y, true.numpy())
    print("f16c6 conv: vs its own three terms in fp64", e_emu, " vs the fp64 conv", e_true)
    assert e_emu <= 5e-5, e_emu
    assert e_true <= 2e-4, e_true


def test_conv_f16c_spade_epilogue_writes_the_f16c6_image(ctx):
    """The ping-pong kernel's LDS-assembled SPADE epilogue with out_mode 5: hi = f16_rn(v); the block scale of a pixel's 32
    channels is the smallest power of two >= max|v| / 7.5; h6 / l6 are e2m3 of v and of v - hi on that scale."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(59)
    B, r, C, shift = 3, 32, 128, 1
    h = torch.relu(torch.randn((B, r, r, 128), generator=g)).cuda()
    wg = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    wb_ = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    bg, bb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    x = (3 + 2 * torch.randn((B, r >> shift, r >> shift, C), generator=g)).cuda()
    mean = x.mean((0, 1, 2)).contiguous()
    std = torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-5).contiguous()
    w, bias = ops.spade_layout(wg, wb_, bg, bb)
    himg, _ = ops.f16c_activation_image(ops.pad_nhwc(h))
    wimg, wexp, _ = ops.f16c_weight_image(w)
    kw = dict(epilogue=ops.EPI_SPADE, aux=x, aux_shift=shift, mean=mean, std=std, out_padded=True)
    y32 = ops.conv3x3_f16c(ctx, himg, wimg, wexp, bias, r, out_mode=0, **kw)
    y6 = ops.conv3x3_f16c(ctx, himg, wimg, wexp, bias, r, out_mode=5, **kw)
    want = y32[:, 1:-1, 1:-1].contiguous()
    ref_img, (rhi, rh6, rl6) = ops.f16c6_activation_image(want)
    hi, h6, l6 = (t[:, 1:-1, 1:-1] for t in ops.f16c6_decode(y6))
    # (the fp32 run divides by sigma where the packed one multiplies by 1 / sigma: the two agree to an ulp, not bit for bit)
    scale = float(want.abs().max())
    assert float((hi - want.double()).abs().max()) <= 2.0 ** -11 * scale               # hi is the fp16 rounding of the value
    blk = want.double().abs().reshape(B, r, r, C // 32, 32).amax(-1, keepdim=True).expand(B, r, r, C // 32, 32).reshape(want.shape)
    assert float(((h6 - want.double()).abs() / blk).max()) <= 0.07                     # e2m3 on the block scale
    assert float(((hi + l6 - want.double()).abs() / blk).max()) <= 2.0 ** -13
    # the device picks the scales of the host restatement (a block whose max / 7.5 sits within an ulp of a power of two may
    # land one binade off: allow 1 %), and the second piece's scale is the first one's / 2^11
    got = y6[:, 1:-1, 1:-1].contiguous().view(torch.uint8).reshape(-1, 128)
    ref = ref_img.contiguous().view(torch.uint8).reshape(-1, 128)
    assert int((got[:, 88] != ref[:, 88]).sum()) <= got.shape[0] // 100
    assert torch.equal(got[:, 88].int() - 11, got[:, 120].int())
    assert int(got[:, 89:96].abs().max()) == 0 and int(got[:, 121:128].abs().max()) == 0
    full = ops.f16c6_decode(y6)
    assert float(full[0][:, 0].abs().max()) == 0 and float(full[1][:, :, -1].abs().max()) == 0     # the border stays zero


@pytest.mark.parametrize("mode", ["0", "2"])
def test_f16c_convs_under_the_other_kernel_dispatch(mode):
    """PREC_F16C launches go to the stream kernel (conv_sw.hip) for the bias / residual epilogues and to the ping-pong kernel
    for the SPADE epilogue by default; MSR_F16C_SW (read once per process) = 0 sends everything to the ping-pong kernel, 2
    everything to the stream kernel.  The f16c kernel tests above must hold under both: run them in a child process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MSR_F16C_SW=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "f16c and not other_kernel_dispatch"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("B,S,r,C,shift", [(8, 128, 32, 1024, 1), (2, 256, 64, 128, 0), (8, 64, 64, 64, 1), (3, 512, 128, 64, 1),
                                           (8, 512, 16, 1024, 1)])
def test_spade_layer_resident_kernel(ctx, B, S, r, C, shift):
    """conv_gb_resident (csrc/conv_gbr.hip): nearest resize + mask-embedding conv + ReLU + gamma|beta conv + SPADE epilogue in
    one launch, against the float64 chain of the same ops (spade.py:17-24, blocks.py:30-34).  The products are f16c6
    (fp16 main term + fp6 cross pieces): <= 2e-4 of the output's range, like the f16c kernel's SPADE test; the written
    image is the consumer's f16c chunk image (hi = fp16 of the value, hi + lo8 ~ 15 bits, h8 = e4m3 of it), border zero."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(1000 * B + r + C)
    src = (torch.rand((B, S, S, 2), generator=g) - 0.5).cuda()
    we = (torch.randn((3, 3, 2, 128), generator=g) / 3).cuda()
    be = (0.1 * torch.randn(128, generator=g)).cuda()
    wg = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    wb_ = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    bg, bb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    x = (3 + 2 * torch.randn((B, r >> shift, r >> shift, C), generator=g)).cuda()
    mean = x.mean((0, 1, 2)).contiguous()
    std = torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-5).contiguous()
    w, bias = ops.spade_layout(wg, wb_, bg, bb)
    wimg = ops.gbr_weight_image(w)
    f = S // r
    mask = src[:, f // 2::f, f // 2::f][:, :r, :r]                       # tf.image.resize(method="nearest"), half-pixel centres
    e = torch.relu(ref_conv(mask, we, be, 1))
    xr = x.double().cpu()
    if shift:
        xr = xr.repeat_interleave(2, 1).repeat_interleave(2, 2)
    v = ref_conv(e, wg, bg, 1) * ((xr - mean.double().cpu()) / std.double().cpu()) + ref_conv(e, wb_, bb, 1)
    want = torch.where(v >= 0, v, 0.2 * v)
    y = ops.spade_gbr(ctx, src, we, be, wimg, bias, r, x, shift, mean, std)
    full = ops.f16c_decode(y)
    hi, h8, lo8 = (t.cpu()[:, 1:-1, 1:-1] for t in full)
    err = rel_linf((hi + lo8).numpy(), want.numpy())
    print(f"spade layer, resident kernel B={B} S={S} r={r} C={C}: rel L-inf vs float64 {err:.3e}")
    assert err <= 2e-4
    scale = float(want.abs().max())
    assert float((hi - want).abs().max()) <= (2.0 ** -11 + 2e-4) * scale
    assert float(((h8 - want).abs() / want.abs().clamp_min(2.0 ** -6)).max()) <= 2.0 ** -4 * 1.01 + 2e-3
    assert float(full[0][:, 0].abs().max()) == 0 and float(full[1][:, :, -1].abs().max()) == 0 and \
        float(full[0][:, -1].abs().max()) == 0 and float(full[2][:, :, 0].abs().max()) == 0        # the border stays zero


def test_f16c_saturation_regimes(ctx):
    """What the f16c path does with activations outside the pieces' ranges (kernels.h msr_store_f16c4_dev):
      |a| <= 448            all three terms live: per-product error ~2^-15 (the parity regime; every activation of the
                            BASELINE shapes with Keras-default weights is inside it: max |a| ~ 40);
      448 < |a| <= 65504    the e4m3 piece h8 saturates at 448 (and, beyond ~900, l8 = e4m3((a - hi) * 2^11) does too), so the
                            cross terms are short: the conv degrades to ~2^-11 per product (the fp16 rounding of weight and
                            activation) — stated here as <= 1e-3 of the output range;
      |a| > 65504           the producer clamps to +-65504: finite and wrong (fp32 would carry on); never an infinity.
    Consumer side (conv_igemm_f16c_sw through msr_op_conv3x3_f16c) on host-built images, producer side (conv_gb_resident's
    epilogue) through a gamma bias that pushes the SPADE output past both limits."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(77)
    B, r, cin, cout = 2, 32, 128, 128
    x0 = torch.randn((B, r, r, cin), generator=g).cuda()
    w = (torch.randn((3, 3, cin, cout), generator=g) / np.sqrt(9 * cin)).cuda()
    b = torch.zeros(cout).cuda()
    wimg, wexp, _ = ops.f16c_weight_image(ops.kernel_layout(w))
    errs = {}
    for name, scale in (("inside", 100.0), ("h8 saturated", 1.0e4)):        # max |a| ~ 4.5 sigma: 450 -> inside needs 448
        x = (x0 * scale / 4.6).clamp(-scale, scale)
        ximg, _ = ops.f16c_activation_image(ops.pad_nhwc(x))
        y = ops.conv3x3_f16c(ctx, ximg, wimg, wexp, b, r).cpu().numpy()
        errs[name] = rel_linf(y, ref_conv(x, w, b, 1).numpy())
    print("f16c conv, activations inside the pieces' range / beyond 448:", errs)
    assert errs["inside"] <= 1e-4 and errs["h8 saturated"] <= 1e-3, errs
    assert errs["h8 saturated"] > errs["inside"]                           # the regime exists: that is what is documented
    # producer: SPADE output = lrelu(gamma * normalised + beta) with gamma bias 3e4 and normalised ~ +-3 -> |a| up to ~9e4
    Bp, S, rr, C = 2, 128, 32, 64
    src = (torch.rand((Bp, S, S, 2), generator=g) - 0.5).cuda()
    we = (torch.randn((3, 3, 2, 128), generator=g) / 3).cuda()
    be = (0.1 * torch.randn(128, generator=g)).cuda()
    wg = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    wb_ = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
    bg, bb = torch.full((C,), 3.0e4).cuda(), torch.randn(C, generator=g).cuda()
    xx = torch.randn((Bp, rr, rr, C), generator=g).cuda()
    mean = xx.mean((0, 1, 2)).contiguous()
    std = torch.sqrt(xx.var((0, 1, 2), unbiased=False) + 1e-5).contiguous()
    wl, bias = ops.spade_layout(wg, wb_, bg, bb)
    y = ops.spade_gbr(ctx, src, we, be, ops.gbr_weight_image(wl), bias, rr, xx, 0, mean, std)
    hi, h8, lo8 = (t.cpu()[:, 1:-1, 1:-1] for t in ops.f16c_decode(y))
    f = S // rr
    e = torch.relu(ref_conv(src[:, f // 2::f, f // 2::f][:, :rr, :rr], we, be, 1))
    v = ref_conv(e, wg, bg, 1) * ((xx.double().cpu() - mean.double().cpu()) / std.double().cpu()) + ref_conv(e, wb_, bb, 1)
    want = torch.where(v >= 0, v, 0.2 * v)
    assert float(want.abs().max()) > 65504 and torch.isfinite(hi).all() and torch.isfinite(h8).all() and torch.isfinite(lo8).all()
    over = want.abs() > 65504
    assert over.any() and float((hi[over].abs() - 65504).abs().max()) == 0      # clamped, with the sign kept
    assert bool((torch.sign(hi[over]) == torch.sign(want[over])).all())
    mid = (want.abs() > 448) & (want.abs() < 6.0e4)
    assert float((h8[mid].abs() - 448).abs().max()) == 0                        # the e4m3 piece saturates at 448
    assert float(((hi + lo8)[mid] - want[mid]).abs().max() / 6.0e4) <= 2.0 ** -11   # hi alone: fp16's 11 bits


def test_head_kernel_known_answers(ctx):
    """Kernel-level KAT of the head on the HIP side (the oracle side: tests/test_oracle_generator.py):
      * delta image -> the 1-before / 2-after SAME padding of Conv2D(1, 4) after UpSampling2D(2) (networks.py:55-56): a one
        at half-resolution pixel (y, x), channel c lights the 2 x 2 up-sampled block, so out[Y][X] = sum of k[kh][kw][c] over
        the taps with 2y <= Y - 1 + kh <= 2y + 1 (and the same in X);
      * random tensors against torch's conv2d on the explicitly up-sampled, (1, 2)-padded input;
      * the pix2pix variant: Conv2DTranspose(1, 4, 2, 'same') + tanh (pix2pix.py:53-57) against conv_transpose2d."""
    from moonsuperresolution_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    B, r, C = 2, 16, 32
    k = torch.randn((4, 4, C), generator=g)
    # delta probes (leaky relu is the identity on the positive delta)
    for (y, x, c) in ((0, 0, 0), (15, 15, 31), (7, 3, 5), (0, 15, 17)):
        xin = torch.zeros((B, r, r, C))
        xin[1, y, x, c] = 1.0
        got = ops.head(ctx, xin.cuda(), k.numpy(), 0.25).cpu()
        want = torch.full((B, 2 * r, 2 * r), 0.25)
        for Y in range(2 * r):
            for X in range(2 * r):
                acc = 0.0
                for kh in range(4):
                    for kw in range(4):
                        if 2 * y <= Y - 1 + kh <= 2 * y + 1 and 2 * x <= X - 1 + kw <= 2 * x + 1:
                            acc += float(k[kh, kw, c])
                want[1, Y, X] += acc
        assert torch.allclose(got, want, atol=1e-6), (y, x, c)
    # random input: up-sample, pad (1 before, 2 after), correlate
    xin = torch.randn((B, r, r, C), generator=g)
    got = ops.head(ctx, xin.cuda(), k.numpy(), -0.1, slope=0.2).cpu()
    a = torch.where(xin >= 0, xin, 0.2 * xin).double().permute(0, 3, 1, 2)
    up = a.repeat_interleave(2, 2).repeat_interleave(2, 3)
    want = F.conv2d(F.pad(up, (1, 2, 1, 2)), k.double().permute(2, 0, 1)[None]).squeeze(1) - 0.1
    assert rel_linf(got.numpy(), want.numpy()) <= 1e-5
    # pix2pix: Conv2DTranspose(1, 4, strides 2, 'same') + tanh; Keras kernel [kh, kw, out = 1, in = C]
    got = ops.head(ctx, xin.cuda(), k.numpy(), 0.05, slope=1.0, transpose_tanh=True).cpu()
    wt = k.double().permute(2, 0, 1)[:, None]                                   # [in, out = 1, kh, kw]
    full = F.conv_transpose2d(xin.double().permute(0, 3, 1, 2), wt, stride=2)       # [B, 1, 2r + 2, 2r + 2]
    want = torch.tanh(full[:, 0, 1:-1, 1:-1] + 0.05)                             # TF 'same': crop 1 on each side
    assert rel_linf(got.numpy(), want.numpy()) <= 1e-5
