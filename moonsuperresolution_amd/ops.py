"""Kernel-level host wrappers (parity tests and micro-benchmarks of single HIP kernels through the C ABI)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

EPI_BIAS, EPI_RES, EPI_SPADE = 0, 1, 2
TILE_128, TILE_64, TILE_128_K16, TILE_HALO, TILE_HALO16, TILE_PP, TILE_FRAG, TILE_F16X2 = 0, 1, 2, 3, 4, 5, 0x40, 0x80


class OpContext:
    """A bare library handle (no weights) for launching single kernels on one device."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("moonsuperresolution_amd needs a HIP device (MI355X / gfx950); there is no CPU fallback")
        self.device = torch.device("cuda", device)
        cfg = _lib.MsrConfig(64, 1, 256, _lib.VARIANT_IDS["gaugan_no_kl"], device, 0)
        self.h = C.c_void_p()
        _lib.raise_for(self.lib, None, self.lib.msr_create(C.byref(cfg), C.byref(self.h)), "msr_create")

    def close(self):
        if self.h:
            self.lib.msr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pad_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[B,r,r,C] -> zero-bordered [B,r+2,r+2,C] (the layout every conv_igemm input uses)."""
    B, r, _, Cc = x.shape
    p = torch.zeros((B, r + 2, r + 2, Cc), dtype=torch.float32, device=x.device)
    p[:, 1:-1, 1:-1] = x
    return p


def kernel_layout(w_hwio: torch.Tensor) -> torch.Tensor:
    """HWIO [3,3,Cin,Cout] -> [9][Cout][Cin]."""
    kh, kw, ci, co = w_hwio.shape
    return w_hwio.reshape(kh * kw, ci, co).permute(0, 2, 1).contiguous()


def spade_layout(wg: torch.Tensor, wb: torch.Tensor, bg: torch.Tensor, bb: torch.Tensor):
    """gamma / beta HWIO kernels -> one [9][2C][Cin] tensor with (32 gamma | 32 beta) interleaved rows + its bias."""
    Cc = wg.shape[3]
    c = torch.arange(Cc, device=wg.device)
    rows_g = (c // 32) * 64 + (c % 32)
    rows_b = rows_g + 32
    kg, kb = kernel_layout(wg), kernel_layout(wb)
    w = torch.empty((9, 2 * Cc, wg.shape[2]), dtype=torch.float32, device=wg.device)
    w[:, rows_g] = kg
    w[:, rows_b] = kb
    bias = torch.empty(2 * Cc, dtype=torch.float32, device=wg.device)
    bias[rows_g] = bg
    bias[rows_b] = bb
    return w.contiguous(), bias


def split_bf16(ctx: OpContext, x: torch.Tensor) -> torch.Tensor:
    """fp32 -> split-bf16 words (same shape, float32 storage)."""
    x = x.contiguous()
    out = torch.empty_like(x)
    rc = ctx.lib.msr_op_split_bf16(ctx.h, x.data_ptr(), out.data_ptr(), x.numel(),
                                   torch.cuda.current_stream(x.device).cuda_stream)
    _lib.raise_for(ctx.lib, ctx.h, rc, "msr_op_split_bf16")
    return out


def split_f16(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> split-fp16 words: every aligned group of 32 values becomes [32 x hi f16 | 32 x lo f16] with
    hi = f16_rn(v), lo = f16_rn(v - hi) (the operand format of the 2-term gamma|beta mode); float32 storage."""
    x = x.contiguous()
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    hl = torch.stack([hi.reshape(-1, 32), lo.reshape(-1, 32)], 1).contiguous()     # [chunks][2][32]
    return hl.view(torch.int16).reshape(-1).view(torch.float32).reshape(x.shape)


def weights_bf16x3(w_kl: torch.Tensor) -> torch.Tensor:
    """Kernel-layout weights [taps][N][Cin] (fp32) -> the MFMA-fragment order conv_igemm_bf16x3 reads:
    [tap][chunk][n-tile][kg][hi|lo][lane = 32*h + j][8 bf16], returned as float32 storage of the same size."""
    taps, N, Cin = w_kl.shape
    hi = w_kl.to(torch.bfloat16)
    lo = (w_kl - hi.float()).to(torch.bfloat16)
    hl = torch.stack([hi, lo], 0)                                        # [2][tap][N][Cin]
    hl = hl.reshape(2, taps, N // 32, 32, Cin // 32, 2, 2, 8)            # [hl][tap][nt][j][cc][kg][h][e]
    hl = hl.permute(1, 4, 2, 5, 0, 6, 3, 7).contiguous()                 # [tap][cc][nt][kg][hl][h][j][e]
    return hl.view(torch.int16).reshape(-1).view(torch.float32).reshape(taps, N, Cin)


def conv3x3(ctx: OpContext, x_padded: torch.Tensor, w_kl: torch.Tensor, bias: torch.Tensor, rout: int, stride: int = 1,
            epilogue: int = EPI_BIAS, aux: Optional[torch.Tensor] = None, aux_shift: int = 0,
            mean: Optional[torch.Tensor] = None, std: Optional[torch.Tensor] = None, out_padded: bool = False,
            tile: int = -1, out: Optional[torch.Tensor] = None, precision: str = "fp32",
            out_split: bool = False) -> torch.Tensor:
    """One conv_igemm launch on torch's current stream.  x_padded [B, rout*stride+2, ., Cin].
    precision="bf16x3": x_padded must hold the split-bf16 chunk image (``split_bf16``); w_kl either the
    split-bf16 image of the kernel layout (``split_bf16(kernel_layout(w))``; tiles 0, 1, 3 = halo and 4 = halo on the 16x16x32 MFMA) or, with
    tile | TILE_FRAG, the fragment-order weights (``weights_bf16x3``; tiles 0 and 1, weights kept in VGPRs)."""
    B, Cin = x_padded.shape[0], x_padded.shape[3]
    N = w_kl.shape[1]
    Cout = N // 2 if epilogue == EPI_SPADE else N
    if out is None:
        shape = (B, rout + 2, rout + 2, Cout) if out_padded else (B, rout, rout, Cout)
        out = (torch.zeros if out_padded else torch.empty)(shape, dtype=torch.float32, device=x_padded.device)
    p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    stream = torch.cuda.current_stream(x_padded.device).cuda_stream
    if precision == "bf16x3":
        rc = ctx.lib.msr_op_conv3x3_bf16x3(ctx.h, x_padded.data_ptr(), w_kl.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                           B, rout, Cin, N, stride, epilogue, p(aux), aux_shift, p(mean), p(std),
                                           1 if out_padded else 0, 1 if out_split else 0, tile, stream)
    else:
        rc = ctx.lib.msr_op_conv3x3(ctx.h, x_padded.data_ptr(), w_kl.data_ptr(), bias.data_ptr(), out.data_ptr(), B,
                                    rout, Cin, N, stride, epilogue, p(aux), aux_shift, p(mean), p(std),
                                    1 if out_padded else 0, tile, stream)
    _lib.raise_for(ctx.lib, ctx.h, rc, "msr_op_conv3x3")
    return out


def fp8_weight_image(w_kl: torch.Tensor):
    """Kernel-layout weights [taps][N][Cin] (fp32) -> (e4m3 bytes [taps][N][Cpad] uint8, wexp [N] int32, dequantised
    fp32 weights) exactly as msr_load_weight quantises them for the fp8 mode: a power-of-two scale per output channel
    that puts the channel's largest |w| into e4m3's top binade."""
    taps, N, Cin = w_kl.shape
    cpad = 128 if Cin <= 128 else (Cin + 255) // 256 * 256
    amax = w_kl.abs().amax(dim=(0, 2)).double().cpu()
    e = torch.zeros(N, dtype=torch.int64)
    nz = amax > 0
    e[nz] = torch.frexp((amax[nz] / 448.0).float())[1].to(torch.int64)
    scale = torch.pow(2.0, e.double()).float().to(w_kl.device)
    q = (w_kl / scale[None, :, None]).to(torch.float8_e4m3fn)
    img = torch.zeros((taps, N, cpad), dtype=torch.uint8, device=w_kl.device)
    img[:, :, :Cin] = q.view(torch.uint8)
    b = (127 + e).to(torch.int64)
    wexp = (b | (b << 8) | (b << 16) | (b << 24)).to(torch.int32).to(w_kl.device)
    return img.contiguous(), wexp.contiguous(), q.float() * scale[None, :, None]


def bf8_activation_image(x_padded: torch.Tensor):
    """[B, r+2, r+2, C] fp32 -> (bf8 e5m2 bytes [B, r+2, r+2, Cpad] uint8, dequantised fp32 values)."""
    C_ = x_padded.shape[-1]
    cpad = 128 if C_ <= 128 else (C_ + 255) // 256 * 256
    q = x_padded.to(torch.float8_e5m2)
    img = torch.zeros(x_padded.shape[:-1] + (cpad,), dtype=torch.uint8, device=x_padded.device)
    img[..., :C_] = q.view(torch.uint8)
    return img.contiguous(), q.float()


def conv3x3_fp8(ctx: OpContext, x_bytes: torch.Tensor, w_bytes: torch.Tensor, wexp: torch.Tensor, bias: torch.Tensor,
                rout: int, epilogue: int = EPI_BIAS, aux: Optional[torch.Tensor] = None, aux_shift: int = 0,
                mean: Optional[torch.Tensor] = None, std: Optional[torch.Tensor] = None, out_padded: bool = False,
                out_mode: int = 0) -> torch.Tensor:
    """One launch of the fp8 form of the persistent ping-pong conv (msr_op_conv3x3_fp8)."""
    B, cpad = x_bytes.shape[0], x_bytes.shape[3]
    N = w_bytes.shape[1]
    Cout = N // 2 if epilogue == EPI_SPADE else N
    if out_mode == 3:
        opad = 128 if Cout <= 128 else (Cout + 255) // 256 * 256
        shape = (B, rout + 2, rout + 2, opad) if out_padded else (B, rout, rout, opad)
        out = torch.zeros(shape, dtype=torch.uint8, device=x_bytes.device)
    else:
        shape = (B, rout + 2, rout + 2, Cout) if out_padded else (B, rout, rout, Cout)
        out = torch.zeros(shape, dtype=torch.float32, device=x_bytes.device)
    p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    rc = ctx.lib.msr_op_conv3x3_fp8(ctx.h, x_bytes.data_ptr(), w_bytes.data_ptr(), wexp.data_ptr(), bias.data_ptr(),
                                    out.data_ptr(), B, rout, cpad, N, epilogue, p(aux), aux_shift, p(mean), p(std),
                                    1 if out_padded else 0, out_mode, torch.cuda.current_stream(x_bytes.device).cuda_stream)
    _lib.raise_for(ctx.lib, ctx.h, rc, "msr_op_conv3x3_fp8")
    return out


def _f16c_pack(hi_f16: torch.Tensor, p_even: torch.Tensor, p_odd: torch.Tensor) -> torch.Tensor:
    """[..., C] f16 + two [..., C] float8 tensors -> the 128-byte chunk image per 32 channels, as float32 storage [..., C]:
    [32 x f16 | 32 bytes of the first float8 tensor | 32 bytes of the second]."""
    shp = hi_f16.shape
    n = shp[-1] // 32
    img = torch.empty(shp[:-1] + (n, 128), dtype=torch.uint8, device=hi_f16.device)
    img[..., 0:64] = hi_f16.contiguous().view(torch.uint8).reshape(shp[:-1] + (n, 64))
    e = p_even.contiguous().view(torch.uint8).reshape(shp[:-1] + (n, 32))
    o = p_odd.contiguous().view(torch.uint8).reshape(shp[:-1] + (n, 32))
    img[..., 64:96], img[..., 96:128] = e, o
    return img.reshape(shp[:-1] + (n * 128,)).view(torch.float32).reshape(shp)


def f16c_activation_image(x: torch.Tensor):
    """fp32 [..., C] (C % 32 == 0) -> (f16c chunk image as float32 storage, (hi, h8, lo8) de-quantised float64 parts):
    hi = f16_rn(v), h8 = e4m3(v), lo8 = e4m3((v - hi) * 2^11) / 2^11 — what the SPADE epilogue / mask embedding write."""
    hi = x.to(torch.float16)
    lo = x - hi.float()
    h8 = x.clamp(-448, 448).to(torch.float8_e4m3fn)
    l8 = (lo * 2048.0).clamp(-448, 448).to(torch.float8_e4m3fn)
    return _f16c_pack(hi, h8, l8), (hi.double(), h8.double(), l8.double() / 2048.0)


def f16c_weight_image(w_kl: torch.Tensor):
    """Kernel-layout weights [taps][N][Cin] -> (f16c image, wexp int32 [N], (hi, h8, lo8) de-quantised): pieces are stored
    lo-first so that piece g of a weight row pairs with piece g of an activation row (w_lo * x_hi, w_hi * x_lo)."""
    hi = w_kl.to(torch.float16)
    lo = w_kl - hi.float()

    def pow2exp(t):
        amax = t.abs().amax(dim=(0, 2)).float().cpu()
        e = torch.zeros(amax.shape, dtype=torch.int64)
        nz = amax > 0
        e[nz] = torch.frexp(amax[nz] / 448.0)[1].to(torch.int64)
        return e
    eh, el = pow2exp(w_kl), pow2exp(lo)
    sh = torch.pow(2.0, eh.double()).float().to(w_kl.device)[None, :, None]
    sl = torch.pow(2.0, el.double()).float().to(w_kl.device)[None, :, None]
    h8 = (w_kl / sh).to(torch.float8_e4m3fn)
    l8 = (lo / sl).to(torch.float8_e4m3fn)
    wexp = ((127 + el) | ((127 + eh) << 8)).to(torch.int32).to(w_kl.device)
    return _f16c_pack(hi, l8, h8), wexp.contiguous(), (hi.double(), h8.double() * sh.double(), l8.double() * sl.double())


# ---- f16c6: fp16 main term + fp6 e2m3 cross pieces with block scales (csrc/kernels.h PREC_F16C6) ------------------------------
_E2M3 = torch.tensor([i / 8 for i in range(8)] + [1 + i / 8 for i in range(8)] + [2 + i / 4 for i in range(8)] +
                     [4 + i / 2 for i in range(8)], dtype=torch.float64)


def _e2m3_codes(q: torch.Tensor) -> torch.Tensor:
    """|q| <= 7.5 (float64) -> 6-bit codes (uint8), round to nearest (ties to the even code, as the hardware converter)."""
    grid = _E2M3.to(q.device)
    a = q.abs().clamp(max=7.5)
    idx = torch.bucketize(a, (grid[1:] + grid[:-1]) / 2, right=False)
    # a tie sits exactly on a midpoint: bucketize(right=False) puts it in the lower bucket; RNE wants the even code
    mid = (grid[1:] + grid[:-1]) / 2
    tie = (idx < 31) & (a == mid[idx.clamp(max=30)])
    idx = torch.where(tie & (idx % 2 == 1), idx + 1, idx)
    return (idx.to(torch.uint8) | ((q < 0) & (a > 0)).to(torch.uint8) * 32).to(torch.uint8)


def _pack6(codes: torch.Tensor) -> torch.Tensor:
    """[..., 32] 6-bit codes -> [..., 24] bytes, little-endian bit string (element c at bits 6c..6c+5)."""
    c = codes.to(torch.int64).reshape(codes.shape[:-1] + (8, 4))
    w = c[..., 0] | (c[..., 1] << 6) | (c[..., 2] << 12) | (c[..., 3] << 18)       # 24 bits per 4 codes
    b = torch.stack([w & 255, (w >> 8) & 255, (w >> 16) & 255], dim=-1)
    return b.reshape(codes.shape[:-1] + (24,)).to(torch.uint8)


def _f16c6_pack(hi: torch.Tensor, first: torch.Tensor, e_first: torch.Tensor, second: torch.Tensor, e_second: torch.Tensor):
    """hi [..., C] fp16, the two code pieces [..., C] and their e8m0 bytes [..., C // 32] -> float32 storage [..., C]."""
    shp = hi.shape
    n = shp[-1] // 32
    out = torch.zeros(shp[:-1] + (n, 128), dtype=torch.uint8, device=hi.device)
    out[..., 0:64] = hi.contiguous().view(torch.uint8).reshape(shp[:-1] + (n, 64))
    out[..., 64:88] = _pack6(first.reshape(shp[:-1] + (n, 32)))
    out[..., 88] = e_first.to(torch.uint8)
    out[..., 96:120] = _pack6(second.reshape(shp[:-1] + (n, 32)))
    out[..., 120] = e_second.to(torch.uint8)
    return out.reshape(shp[:-1] + (n * 128,)).view(torch.float32).reshape(shp)


def _ceil_log2_over(amax: torch.Tensor, top: float) -> torch.Tensor:
    """E = ceil(log2(amax / top)) as int64 (0 where amax == 0)"""
    m, e = torch.frexp((amax.double() / top))
    E = torch.where(m == 0.5, e - 1, e).to(torch.int64)
    return torch.where(amax > 0, E, torch.zeros_like(E))


def f16c6_activation_image(x: torch.Tensor):
    """fp32 [..., C] -> (f16c6 image, (hi, h6, l6) de-quantised float64 parts): one scale 2^E >= max|x| / 7.5 per pixel and
    32-channel chunk, h6 = e2m3(x / 2^E), l6 = e2m3((x - hi) / 2^(E - 11)) — what the SPADE epilogue writes."""
    shp = x.shape
    n = shp[-1] // 32
    hi = x.to(torch.float16)
    lo = (x - hi.float()).double()
    xb = x.double().reshape(shp[:-1] + (n, 32))
    E = _ceil_log2_over(xb.abs().amax(-1), 7.5).clamp(-100, 120)
    s = torch.pow(2.0, E.double())[..., None]
    ch = _e2m3_codes(xb / s)
    cl = _e2m3_codes(lo.reshape(xb.shape) / (s / 2048.0))
    dec = lambda c: torch.where(c >= 32, -_E2M3.to(c.device)[(c & 31).long()], _E2M3.to(c.device)[(c & 31).long()])   # noqa: E731
    img = _f16c6_pack(hi, ch, 127 + E, cl, 127 + E - 11)
    return img, (hi.double(), (dec(ch) * s).reshape(shp), (dec(cl) * s / 2048.0).reshape(shp))


def f16c6_weight_image(w_kl: torch.Tensor):
    """Kernel-layout weights [taps][N][Cin] -> (f16c6 image, (hi, h6, l6) de-quantised): one scale per output channel and piece;
    the lo piece is stored first so that half g of a weight row pairs with half g of an activation row."""
    hi = w_kl.to(torch.float16)
    lo = (w_kl - hi.float()).double()
    Eh = _ceil_log2_over(w_kl.abs().amax(dim=(0, 2)), 7.5).clamp(-100, 100)
    El = _ceil_log2_over(lo.abs().amax(dim=(0, 2)), 7.5).clamp(-100, 100)
    sh = torch.pow(2.0, Eh.double())[None, :, None]
    sl = torch.pow(2.0, El.double())[None, :, None]
    ch, cl = _e2m3_codes(w_kl.double() / sh), _e2m3_codes(lo / sl)
    n = w_kl.shape[-1] // 32
    eb = lambda E: (127 + E)[None, :, None].expand(w_kl.shape[0], -1, n)   # noqa: E731
    dec = lambda c: torch.where(c >= 32, -_E2M3.to(c.device)[(c & 31).long()], _E2M3.to(c.device)[(c & 31).long()])   # noqa: E731
    img = _f16c6_pack(hi, cl, eb(El), ch, eb(Eh))
    return img, (hi.double(), dec(ch) * sh, dec(cl) * sl)


def f16c6_decode(img: torch.Tensor):
    """f16c6 activation image -> (hi, h6, l6) float64 tensors [..., C] (pieces multiplied by their block scales)."""
    shp = img.shape
    n = shp[-1] // 32
    b = img.contiguous().view(torch.uint8).reshape(shp[:-1] + (n, 128)).long()
    hi = img.contiguous().view(torch.uint8).reshape(shp[:-1] + (n, 128))[..., 0:64].contiguous().view(torch.float16).reshape(shp).double()

    def piece(off):
        by = b[..., off:off + 24].reshape(shp[:-1] + (n, 8, 3))
        w = by[..., 0] | (by[..., 1] << 8) | (by[..., 2] << 16)
        c = torch.stack([(w >> (6 * k)) & 63 for k in range(4)], dim=-1).reshape(shp[:-1] + (n, 32))
        grid = _E2M3.to(img.device)
        v = torch.where(c >= 32, -grid[c & 31], grid[c & 31])
        s = torch.pow(2.0, (b[..., off + 24] - 127).double())[..., None]
        return (v * s).reshape(shp)
    return hi, piece(64), piece(96)


def conv3x3_f16c(ctx: OpContext, x_img: torch.Tensor, w_img: torch.Tensor, wexp: torch.Tensor, bias: torch.Tensor, rout: int,
                 epilogue: int = EPI_BIAS, aux: Optional[torch.Tensor] = None, aux_shift: int = 0,
                 mean: Optional[torch.Tensor] = None, std: Optional[torch.Tensor] = None, out_padded: bool = False,
                 out_mode: int = 0) -> torch.Tensor:
    """One launch of the f16c conv (msr_op_conv3x3_f16c; the library sends it to the ping-pong or the stream kernel).
    ``wexp=None``: the operands are f16c6 images (fp6 pieces, stream kernel)."""
    B, Cin = x_img.shape[0], x_img.shape[3]
    N = w_img.shape[1]
    Cout = N // 2 if epilogue == EPI_SPADE else N
    shape = (B, rout + 2, rout + 2, Cout) if out_padded else (B, rout, rout, Cout)
    out = torch.zeros(shape, dtype=torch.float32, device=x_img.device)
    p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    rc = ctx.lib.msr_op_conv3x3_f16c(ctx.h, x_img.data_ptr(), w_img.data_ptr(), p(wexp), bias.data_ptr(),
                                     out.data_ptr(), B, rout, Cin, N, epilogue, p(aux), aux_shift, p(mean), p(std),
                                     1 if out_padded else 0, out_mode, torch.cuda.current_stream(x_img.device).cuda_stream)
    _lib.raise_for(ctx.lib, ctx.h, rc, "msr_op_conv3x3_f16c")
    return out


GBR_PERM = [8 * (e >> 3) + 4 * (e & 1) + ((e >> 1) & 3) for e in range(32)]   # csrc/conv_gbr.hip: position e <- channel


def gbr_weight_image(w_kl: torch.Tensor) -> torch.Tensor:
    """gamma|beta weights [9][N][128] (kernel layout, spade_layout rows) -> the weight STREAM conv_gb_resident reads: the f16c6
    image with the input channels of every 32-chunk in the kernel's position order, re-ordered into the order its waves load
    it: [channel block nt][wave q][tap pair P][column block j][piece][lane] x 16 bytes (csrc/conv_gbr.hip, api.hip
    gbr_weight_stream).  Returned as float32 storage [9][N][128] (same byte count)."""
    N = w_kl.shape[1]
    idx = torch.tensor([32 * c + GBR_PERM[e] for c in range(w_kl.shape[2] // 32) for e in range(32)], device=w_kl.device)
    img = f16c6_weight_image(w_kl[:, :, idx].contiguous())[0]
    rec = img.contiguous().view(torch.uint8).reshape(9, N, 4, 128)                 # [tap][row][chunk][128 bytes]
    dev = w_kl.device
    nt, q, P, j, piece, lane = torch.meshgrid(torch.arange(N // 128), torch.arange(4), torch.arange(18), torch.arange(2),
                                              torch.arange(4), torch.arange(64), indexing="ij")
    px, cg = lane & 15, lane >> 4
    row = 128 * nt + 64 * (q >> 1) + 16 * (q & 1) + 32 * j + px
    T = torch.where(piece < 2, 2 * P + piece, 2 * P + (cg >> 1))
    off = torch.where(piece < 2, 16 * cg, 64 + 32 * (cg & 1) + 16 * (piece - 2))
    tap, chunk = (T % 9).to(dev), (T // 9).to(dev)
    byte = off.to(dev)[..., None] + torch.arange(16, device=dev)
    out = rec[tap[..., None], row.to(dev)[..., None], chunk[..., None], byte]        # [..., 16] bytes
    return out.contiguous().reshape(-1).view(torch.float32).reshape(9, N, 128)


def spade_gbr(ctx: OpContext, src: torch.Tensor, we: torch.Tensor, be: torch.Tensor, w_img: torch.Tensor, bias: torch.Tensor,
              r: int, x: torch.Tensor, aux_shift: int, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """One launch of conv_gb_resident (msr_op_spade_gbr): resize + mask embedding + gamma|beta conv + SPADE epilogue;
    returns the zero-bordered f16c image [B, r + 2, r + 2, C] (float32 storage; f16c_decode reads it)."""
    B, S = src.shape[0], src.shape[1]
    N = w_img.shape[1]
    out = torch.zeros((B, r + 2, r + 2, N // 2), dtype=torch.float32, device=src.device)
    rc = ctx.lib.msr_op_spade_gbr(ctx.h, src.data_ptr(), S, we.data_ptr(), be.data_ptr(), w_img.data_ptr(), bias.data_ptr(),
                                  out.data_ptr(), B, r, N, x.data_ptr(), aux_shift, mean.data_ptr(), std.data_ptr(),
                                  torch.cuda.current_stream(src.device).cuda_stream)
    _lib.raise_for(ctx.lib, ctx.h, rc, "msr_op_spade_gbr")
    return out


def f16c_decode(img: torch.Tensor):
    """f16c activation image (float32 storage [..., C]) -> (hi, h8, lo8) as float64 tensors [..., C]."""
    shp = img.shape
    n = shp[-1] // 32
    b = img.contiguous().view(torch.uint8).reshape(shp[:-1] + (n, 128))
    hi = b[..., 0:64].contiguous().view(torch.float16).reshape(shp).double()
    h8 = b[..., 64:96].contiguous().view(torch.float8_e4m3fn).reshape(shp).double()
    l8 = b[..., 96:128].contiguous().view(torch.float8_e4m3fn).reshape(shp).double()
    return hi, h8, l8 / 2048.0


def head(ctx: OpContext, x: torch.Tensor, kernel: "np.ndarray", bias: float, slope: float = 0.2, transpose_tanh: bool = False) -> torch.Tensor:
    """One launch of the head kernel (msr_op_head): x dense [B, r, r, C] -> [B, 2r, 2r].  kernel [4, 4, C] float32 (host):
    the Conv2D(1, 4, 'same') kernel applied after UpSampling2D(2) (networks.py:54-56), or with ``transpose_tanh`` the
    Conv2DTranspose(1, 4, 2, 'same') kernel followed by tanh (pix2pix.py:53-57)."""
    import numpy as np
    B, r, _, Cc = x.shape
    k = np.ascontiguousarray(kernel, dtype=np.float32)
    assert k.shape == (4, 4, Cc)
    out = torch.empty((B, 2 * r, 2 * r), dtype=torch.float32, device=x.device)
    rc = ctx.lib.msr_op_head(ctx.h, x.contiguous().data_ptr(), k.ctypes.data_as(C.c_void_p), float(bias), out.data_ptr(), B, r, Cc,
                             float(slope), 1 if transpose_tanh else 0, torch.cuda.current_stream(x.device).cuda_stream)
    _lib.raise_for(ctx.lib, ctx.h, rc, "msr_op_head")
    return out
