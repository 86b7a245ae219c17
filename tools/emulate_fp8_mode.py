"""Diagnostic (not a pytest, CPU only): end-to-end error of single-product fp8 convolutions (BASELINE configs[4]) under
different operand recipes, emulated inside the oracle on the layers the GPU fp8 mode covers (3 x 3 stride-1 convs of
rb3..rb6: r >= S / 16).   usage: python tools/emulate_fp8_mode.py [S] [B]

recipes (activation | weight):
  shipped_r2   e5m2, no scale                      | e4m3, one power of two per output channel
  e4m3_blk     e4m3, 2^E per (pixel, 32 channels)  | e4m3, one power of two per output channel
  e4m3_blk_wb  e4m3, 2^E per (pixel, 32 channels)  | e4m3, 2^E per (output channel, tap, 32 input channels)
  *_gb / *_main: the recipe on the gamma|beta convs only / on the main convs only (the other kind exact)
"""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import generator_ref as G
from moonsuperresolution_amd import make_latent_noise, make_weights, synthetic_patches

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
DT = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32


def e4m3(v):
    return v.float().to(torch.float8_e4m3fn).to(v.dtype)


def e5m2(v):
    return v.float().to(torch.float8_e5m2).to(v.dtype)


def pow2_scale(amax, mx):
    return torch.exp2(torch.ceil(torch.log2(amax.clamp(min=1e-30) / mx)))


def make_conv(act, wgt, which):
    def conv(x, k, bias=None, stride=1):
        if not (k.shape[0] == 3 and stride == 1 and k.shape[2] >= 128 and x.shape[1] >= S // 16):
            return ORIG(x, k, bias, stride)
        is_gb = k.shape[2] == 128 and bias is None or False
        kind = "gb" if k.shape[2] == 128 else "main"
        if which != "all" and which != kind:
            return ORIG(x, k, bias, stride)
        Bn, H, W, C = x.shape
        if act == "e5m2":
            qx = e5m2(x)
        else:
            xb = x.reshape(Bn, H, W, C // 32, 32)
            sc = pow2_scale(xb.abs().amax(-1, keepdim=True), 448.0)
            qx = (e4m3(xb / sc) * sc).reshape(x.shape)
        if wgt == "chan":
            sc = pow2_scale(k.abs().amax((0, 1, 2), keepdim=True), 448.0)
            qk = e4m3(k / sc) * sc
        else:
            kh, kw, ci, co = k.shape
            kb = k.reshape(kh, kw, ci // 32, 32, co)
            sc = pow2_scale(kb.abs().amax(3, keepdim=True), 448.0)
            qk = (e4m3(kb / sc) * sc).reshape(k.shape)
        return ORIG(qx, qk, bias, 1)
    return conv


ORIG = G.conv2d_same
w = make_weights("gaugan", S, seed=1234)
eps = make_latent_noise(B, 256, 7)
x = synthetic_patches(B, S, seed=0)
ref = G.spade_call(x, w, "gaugan", eps=eps, dtype=DT)
span = np.abs(ref).max()
for name, act, wgt, which in (("shipped_r2", "e5m2", "chan", "all"), ("e4m3_blk", "e4m3", "chan", "all"),
                              ("e4m3_blk_wb", "e4m3", "blk", "all"), ("e4m3_blk_gb", "e4m3", "chan", "gb"),
                              ("e4m3_blk_main", "e4m3", "chan", "main"), ("shipped_r2_gb", "e5m2", "chan", "gb"),
                              ("shipped_r2_main", "e5m2", "chan", "main")):
    G.conv2d_same = make_conv(act, wgt, which)
    y = G.spade_call(x, w, "gaugan", eps=eps, dtype=DT)
    G.conv2d_same = ORIG
    d = y - ref
    print(f"{name:16s} rel L-inf {np.abs(d).max() / span:.3e}  rel rms {np.sqrt((d ** 2).mean()) / np.sqrt((ref ** 2).mean()):.3e}", flush=True)
