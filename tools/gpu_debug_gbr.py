"""Diagnostic (not a pytest): where conv_gb_resident (csrc/conv_gbr.hip) differs from the float64 chain.
Probes: beta = one tap of one embedding channel copied to an output channel (gamma = 0, x = mean): isolates phase 1, the
channel order, the tap addressing and the epilogue's channel / pixel mapping.   usage: python tools/gpu_debug_gbr.py"""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from moonsuperresolution_amd import ops

ctx = ops.OpContext()
B, S, r, C, shift = 8, 64, 64, 64, 0
g = torch.Generator(device="cpu").manual_seed(5)
src = (torch.rand((B, S, S, 2), generator=g) - 0.5).cuda()
we = (torch.randn((3, 3, 2, 128), generator=g) / 3).cuda()
be = (0.1 * torch.randn(128, generator=g)).cuda()


def ref_conv(x, w, b):
    x, w, b = x.double().cpu(), w.double().cpu(), b.double().cpu()
    return F.conv2d(F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1)), w.permute(3, 2, 0, 1), b).permute(0, 2, 3, 1)


f = S // r
mask = src[:, f // 2::f, f // 2::f][:, :r, :r]
E = torch.relu(ref_conv(mask, we, be))          # [B, r, r, 128] float64


def run(wg, wb_, bg, bb, x, mean, std):
    w, bias = ops.spade_layout(wg, wb_, bg, bb)
    y = ops.spade_gbr(ctx, src, we, be, ops.gbr_weight_image(w), bias, r, x, shift, mean, std)
    hi, h8, lo8 = (t.cpu()[:, 1:-1, 1:-1] for t in ops.f16c_decode(y))
    return hi + lo8


def report(name, got, want):
    d = (got - want).abs()
    sc = float(want.abs().max())
    print(f"{name}: rel L-inf {float(d.max()) / sc:.3e}")
    if float(d.max()) / sc > 1e-3:
        bad = d > 1e-3 * sc
        print("   bad fraction", float(bad.double().mean()), " per batch", [round(float(bad[b].double().mean()), 3) for b in range(B)])
        print("   bad per output channel (first 32):", [round(float(bad[..., c].double().mean()), 2) for c in range(min(32, C))])
        ys = bad.any(-1).any(0)
        print("   bad rows (y) :", "".join("X" if ys[y].any() else "." for y in range(r)))
        print("   bad cols (x) :", "".join("X" if ys[:, x].any() else "." for x in range(r)))
        b, y, x, c = [int(v) for v in torch.nonzero(bad)[0]]
        print(f"   first bad: b={b} y={y} x={x} c={c}: got {float(got[b, y, x, c]):.6f} want {float(want[b, y, x, c]):.6f}")


zeros_x = torch.zeros((B, r, r, C)).cuda()
mean0, std1 = torch.zeros(C).cuda(), torch.ones(C).cuda()
zC = torch.zeros(C).cuda()
# probe 1: beta[c] = E[centre tap, channel c] for c < 64 (embedding channels 0..63), then channels 64..127
for base_ch in (0, 64):
    wb_ = torch.zeros((3, 3, 128, C)).cuda()
    for c in range(C):
        wb_[1, 1, base_ch + c, c] = 1.0
    got = run(torch.zeros((3, 3, 128, C)).cuda(), wb_, zC, zC, zeros_x, mean0, std1)
    report(f"probe centre tap, embedding channels {base_ch}..{base_ch + C - 1}", got, E[..., base_ch:base_ch + C])
    # which embedding channel does output channel c show?
    Ef = E.reshape(-1, 128)
    Gf = got.reshape(-1, C).double()
    amap = []
    for c in range(C):
        d = ((Ef - Gf[:, c:c + 1]) ** 2).sum(0)
        k = int(d.argmin())
        amap.append(k if float(d[k]) < 1e-3 * float((Gf[:, c] ** 2).sum() + 1e-9) else -1)
    print("   output channel c shows embedding channel:", amap)
# probe 1b: output channel c selects embedding channel (c + 5) % 32 of its chunk: does the collapse follow the weights (row
# addressing) or the activations (all positions equal)?
wb_ = torch.zeros((3, 3, 128, C)).cuda()
for c in range(C):
    wb_[1, 1, (c & ~31) + (c + 5) % 32, c] = 1.0
got = run(torch.zeros((3, 3, 128, C)).cuda(), wb_, zC, zC, zeros_x, mean0, std1)
Ef = E.reshape(-1, 128)
Gf = got.reshape(-1, C).double()
amap = []
for c in range(C):
    d = ((Ef - Gf[:, c:c + 1]) ** 2).sum(0)
    k = int(d.argmin())
    amap.append(k if float(d[k]) < 1e-3 * float((Gf[:, c] ** 2).sum() + 1e-9) else -1)
print("probe 1b (c -> (c + 5) % 32): output channel c shows embedding channel:", amap)
# probe 2: single taps of embedding channel c -> output channel c
for (ky, kx) in ((0, 0), (0, 2), (1, 0), (2, 1), (2, 2)):
    wb_ = torch.zeros((3, 3, 128, C)).cuda()
    for c in range(C):
        wb_[ky, kx, c, c] = 1.0
    got = run(torch.zeros((3, 3, 128, C)).cuda(), wb_, zC, zC, zeros_x, mean0, std1)
    want = ref_conv(E.float(), wb_, zC)
    report(f"probe tap ({ky},{kx})", got, want)
# probe 3: gamma path and the normalisation: gamma = E centre, beta = 0, x random
x = (3 + 2 * torch.randn((B, r, r, C), generator=g)).cuda()
mean = x.mean((0, 1, 2)).contiguous()
std = torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-5).contiguous()
wgm = torch.zeros((3, 3, 128, C)).cuda()
for c in range(C):
    wgm[1, 1, c, c] = 1.0
got = run(wgm, torch.zeros((3, 3, 128, C)).cuda(), zC, zC, x, mean, std)
v = E[..., :C] * ((x.double().cpu() - mean.double().cpu()) / std.double().cpu())
report("probe gamma * normalised", got, torch.where(v >= 0, v, 0.2 * v))
# probe 4: random weights (the full test)
wg = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
wb_ = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
bg, bb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
got = run(wg, wb_, bg, bb, x, mean, std)
v = ref_conv(E.float(), wg, bg) * ((x.double().cpu() - mean.double().cpu()) / std.double().cpu()) + ref_conv(E.float(), wb_, bb)
report("random weights", got, torch.where(v >= 0, v, 0.2 * v))

# which of the three terms does the kernel compute?  (E quantised like phase 1: per pixel and 32-channel chunk)
_, (xh, x6, xl) = ops.f16c6_activation_image(E.float().cuda())
wk, _ = ops.spade_layout(wg, wb_, bg, bb)
_, (wh, w6, wl) = ops.f16c6_weight_image(wk)


def conv_k(xq, wq):          # xq [B,r,r,128] float64, wq [9][N][128] float64 (interleaved rows) -> [B,r,r,N]
    w4 = wq.reshape(3, 3, wq.shape[1], 128).permute(2, 3, 0, 1).cpu()
    return F.conv2d(F.pad(xq.cpu().permute(0, 3, 1, 2), (1, 1, 1, 1)), w4).permute(0, 2, 3, 1)


def spade_of(gb):
    cidx = torch.arange(C)
    rows_g = (cidx // 32) * 64 + (cidx % 32)
    gam = gb[..., rows_g] + bg.double().cpu()
    bet = gb[..., rows_g + 32] + bb.double().cpu()
    v = gam * ((x.double().cpu() - mean.double().cpu()) / std.double().cpu()) + bet
    return torch.where(v >= 0, v, 0.2 * v)


main = conv_k(xh, wh)
c1 = conv_k(x6, wl)
c2 = conv_k(xl, wh if False else w6)
for name, gb in (("main only", main), ("main + x_hi*w_lo", main + c1), ("main + x_lo*w_hi", main + c2), ("all three", main + c1 + c2)):
    want = spade_of(gb)
    print(f"kernel vs [{name}]: rel L-inf {float((got - want).abs().max() / want.abs().max()):.3e}")

print("per tap: kernel vs emulation with [main, +x_hi*w_lo, +x_lo*w_hi, all] (beta path only, gamma = 0, x = mean)")
for tap in range(9):
    for chunk in (None,):
        wg0 = torch.zeros((3, 3, 128, C)).cuda()
        wbt = torch.zeros((3, 3, 128, C)).cuda()
        wbt[tap // 3, tap % 3] = (torch.randn((128, C), generator=g) / 12).cuda()
        bbp = torch.full((C,), 4.0).cuda()             # keeps beta positive: no leaky-relu kink
        wk, bias_k = ops.spade_layout(wg0, wbt, zC, bbp)
        _, (wh, w6, wl) = ops.f16c6_weight_image(wk)
        got = run(wg0, wbt, zC, bbp, zeros_x, mean0, std1)
        cidx = torch.arange(C)
        rows_b = (cidx // 32) * 64 + (cidx % 32) + 32
        main = conv_k(xh, wh)[..., rows_b]
        c1 = conv_k(x6, wl)[..., rows_b]
        c2 = conv_k(xl, w6)[..., rows_b]
        res = []
        for gb in (main, main + c1, main + c2, main + c1 + c2):
            want = gb + 4.0
            res.append(float((got - want).abs().max()))
        print(f"  tap {tap}: " + "  ".join(f"{v:.2e}" for v in res), " (|cross| max", f"{float((c1 + c2).abs().max()):.2e})")
        if tap in (0, 4):
            d = (got - (main + c1 + c2 + 4.0))
            dm = (got - (main + 4.0))
            print(f"     rms: vs all {float((d ** 2).mean().sqrt()):.2e}  vs main {float((dm ** 2).mean().sqrt()):.2e}  cross rms {float(((c1 + c2) ** 2).mean().sqrt()):.2e}")
            ad = d.abs()
            thr = 0.5 * float(ad.max())
            bad = ad > thr
            idx = torch.nonzero(bad)
            print("     worst errors at (b, y, x, c):", [tuple(int(v) for v in r_) for r_ in idx[:12]])
            print("     bad count by y % 16:", [int(bad[:, yy::16].sum()) for yy in range(16)])
            print("     bad count by x % 16:", [int(bad[:, :, xx::16].sum()) for xx in range(16)])
            print("     bad count by c % 16:", [int(bad[..., cc::16].sum()) for cc in range(16)])
            # regress d on c1 and c2: d ~ a c1 + b c2
            A = torch.stack([c1.flatten(), c2.flatten()], 1)
            sol = torch.linalg.lstsq(A, dm.flatten().unsqueeze(1)).solution.flatten()
            print(f"     least squares (got - main) ~ {float(sol[0]):.3f} * x_hi*w_lo + {float(sol[1]):.3f} * x_lo*w_hi")

print("isolating the cross terms by zeroing pieces of the weight image (tap 4 only, beta path):")
wg0 = torch.zeros((3, 3, 128, C)).cuda()
wbt = torch.zeros((3, 3, 128, C)).cuda()
wbt[1, 1] = (torch.randn((128, C), generator=g) / 12).cuda()
bbp = torch.full((C,), 4.0).cuda()
wk, bias_k = ops.spade_layout(wg0, wbt, zC, bbp)
_, (wh, w6, wl) = ops.f16c6_weight_image(wk)
cidx = torch.arange(C)
rows_b = (cidx // 32) * 64 + (cidx % 32) + 32
main = conv_k(xh, wh)[..., rows_b]
c1 = conv_k(x6, wl)[..., rows_b]
c2 = conv_k(xl, w6)[..., rows_b]
img = ops.gbr_weight_image(wk)
for name, lo_b, hi_b, want in (("l6 pieces zeroed (expect main + x_lo*w_hi)", 64, 88, main + c2), ("h6 pieces zeroed (expect main + x_hi*w_lo)", 96, 120, main + c1),
                               ("both zeroed (expect main)", 64, 120, main)):
    im = img.clone().contiguous()
    by = im.view(torch.uint8).reshape(9, im.shape[1], 4, 128)
    if lo_b == 64 and hi_b == 120:
        by[..., 64:88] = 0
        by[..., 96:120] = 0
    else:
        by[..., lo_b:hi_b] = 0
    y = ops.spade_gbr(ctx, src, we, be, im, bias_k, r, zeros_x, shift, mean0, std1)
    hi_, h8_, lo8_ = (t.cpu()[:, 1:-1, 1:-1] for t in ops.f16c_decode(y))
    got = hi_ + lo8_
    d = got - (want + 4.0)
    print(f"  {name}: max {float(d.abs().max()):.2e} rms {float((d ** 2).mean().sqrt()):.2e}   (rms of c1 {float((c1 ** 2).mean().sqrt()):.2e}, c2 {float((c2 ** 2).mean().sqrt()):.2e})")
