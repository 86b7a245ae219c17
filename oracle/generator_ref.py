"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement, in PyTorch-CPU, of the reference's generator forward passes:

* ``GauGAN.call``        spade/models/model.py:564-567
* ``GauGAN_no_KL.call``  spade/models/model.py:265-267
* ``CNNSpade.call``      spade/models/model.py:789-791
* ``Pix2Pix().generator``  pix2pix.py:88-108

PARITY UNPINNED: the reference has no tests, golden vectors or weights, and its arithmetic lives
in un-vendored third-party packages that are not installable here (tensorflow-gpu==2.5.0,
tensorflow-addons==0.16.1, keras-nightly==2.5.0.dev2021032900; pip-env.py:15,34,36).  This file
restates the *published* semantics of those ops and is anchored on
the reference's own call sites, cited per function below.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import it.

All tensors are NHWC like the reference; conv kernels are HWIO, transposed-conv kernels are
``[kh, kw, Cout, Cin]``, dense kernels ``[in, out]`` (names: moonsuperresolution_amd/weights.py).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

LEAK = 0.2          # alpha passed everywhere in model.py:362-379 / networks.py:55
SPADE_EPS = 1e-5    # spade.py:6
IN_EPS = 1e-3       # tensorflow_addons InstanceNormalization default epsilon (blocks.py:63)
BN_EPS = 1e-3       # keras BatchNormalization default epsilon (pix2pix.py:71,82)
P2P_LEAK = 0.3      # keras LeakyReLU() default alpha (pix2pix.py:72)


def _t(a, dtype) -> torch.Tensor:
    return torch.as_tensor(np.asarray(a)).to(dtype)


def same_padding(size: int, k: int, stride: int):
    """TF 'SAME' padding (before, after) along one axis; the extra pixel goes AFTER."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return total // 2, total - total // 2


def conv2d_same(x: torch.Tensor, kernel_hwio: torch.Tensor, bias: Optional[torch.Tensor] = None,
                stride: int = 1) -> torch.Tensor:
    """keras.layers.Conv2D(padding='same') on an NHWC tensor: cross-correlation, HWIO kernel."""
    kh, kw = kernel_hwio.shape[:2]
    pt, pb = same_padding(x.shape[1], kh, stride)
    pl, pr = same_padding(x.shape[2], kw, stride)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, kernel_hwio.permute(3, 2, 0, 1), bias, stride=stride)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose_same_s2(x: torch.Tensor, kernel_hwoi: torch.Tensor,
                             bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """keras Conv2DTranspose(k=4, strides=2, padding='same'): out = 2*in, kernel [kh,kw,Cout,Cin].

    Equivalent to torch ConvTranspose2d(k=4, s=2, padding=1).
    """
    assert kernel_hwoi.shape[0] == 4 and kernel_hwoi.shape[1] == 4
    w = kernel_hwoi.permute(3, 2, 0, 1)  # torch wants [Cin, Cout, kh, kw]
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2), w, bias, stride=2, padding=1)
    return y.permute(0, 2, 3, 1)


def leaky_relu(x: torch.Tensor, alpha: float) -> torch.Tensor:
    return torch.where(x >= 0, x, x * alpha)


def resize_nearest_halfpixel(src: torch.Tensor, out_hw: int) -> torch.Tensor:
    """tf.image.resize(method='nearest') of TF2: src index = floor((dst + 0.5) * in / out)."""
    n_in = src.shape[1]
    idx = torch.clamp(torch.floor((torch.arange(out_hw, dtype=torch.float64) + 0.5) * (n_in / out_hw)).long(),
                      max=n_in - 1)
    return src[:, idx][:, :, idx]


def upsample2x(x: torch.Tensor) -> torch.Tensor:
    """keras UpSampling2D((2,2)), nearest: out[i, j] = in[i // 2, j // 2] (networks.py:44-54)."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def spade(x: torch.Tensor, source: torch.Tensor, w: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """SPADE.call, spade.py:16-25 — BATCH moments over (N,H,W), gamma*x_hat + beta (no 1+gamma)."""
    mask = resize_nearest_halfpixel(source, x.shape[1])
    h = torch.relu(conv2d_same(mask, w[f"{prefix}.conv.kernel"], w[f"{prefix}.conv.bias"]))
    gamma = conv2d_same(h, w[f"{prefix}.conv_gamma.kernel"], w[f"{prefix}.conv_gamma.bias"])
    beta = conv2d_same(h, w[f"{prefix}.conv_beta.kernel"], w[f"{prefix}.conv_beta.bias"])
    mean = x.mean(dim=(0, 1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2), keepdim=True)   # tf.nn.moments: biased
    normalized = (x - mean) / torch.sqrt(var + SPADE_EPS)
    return gamma * normalized + beta


def residual_block(x: torch.Tensor, source: torch.Tensor, w, prefix: str, capture=None) -> torch.Tensor:
    """ResidualBlock.call, blocks.py:28-38."""
    y = conv2d_same(leaky_relu(spade(x, source, w, f"{prefix}.spade_1"), LEAK),
                    w[f"{prefix}.conv_1.kernel"], w[f"{prefix}.conv_1.bias"])
    if capture is not None:
        capture[f"{prefix}.x1"] = y.numpy().copy()
    y = conv2d_same(leaky_relu(spade(y, source, w, f"{prefix}.spade_2"), LEAK),
                    w[f"{prefix}.conv_2.kernel"], w[f"{prefix}.conv_2.bias"])
    if f"{prefix}.conv_3.kernel" in w:   # learned skip when filters != input filters (blocks.py:23-26)
        skip = conv2d_same(leaky_relu(spade(x, source, w, f"{prefix}.spade_3"), LEAK),
                           w[f"{prefix}.conv_3.kernel"], w[f"{prefix}.conv_3.bias"])
    else:
        skip = x
    return skip + y


def instance_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """tfa.layers.InstanceNormalization: per sample and channel over H,W, biased var, eps 1e-3."""
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    return (x - mean) / torch.sqrt(var + IN_EPS) * gamma + beta


def encoder(source: torch.Tensor, w, capture=None):
    """build_encoder, networks.py:8-34 with downsample_block, blocks.py:41-68."""
    x = source
    i = 1
    while f"enc.ds{i}.kernel" in w:
        x = conv2d_same(x, w[f"enc.ds{i}.kernel"], None, stride=2)
        if f"enc.ds{i}.in.gamma" in w:
            x = instance_norm(x, w[f"enc.ds{i}.in.gamma"], w[f"enc.ds{i}.in.beta"])
        x = leaky_relu(x, LEAK)
        if capture is not None:
            capture[f"enc.ds{i}.out"] = x.numpy().copy()
        i += 1
    flat = x.reshape(x.shape[0], -1)   # keras Flatten on NHWC
    mean = flat @ w["enc.mean.kernel"] + w["enc.mean.bias"]
    variance = flat @ w["enc.variance.kernel"] + w["enc.variance.bias"]
    return mean, variance


def generator(latent: torch.Tensor, source: torch.Tensor, w, capture=None) -> torch.Tensor:
    """build_generator, networks.py:37-57.  No tanh: the head is leaky_relu -> Conv2D(1, 4, 'same')."""
    sw = source.shape[1] // 64
    x = (latent @ w["gen.dense.kernel"] + w["gen.dense.bias"]).reshape(-1, sw, sw, 1024)
    if capture is not None:
        capture["gen.x0"] = x.numpy().copy()
    i = 1
    while f"gen.rb{i}.conv_1.kernel" in w:
        x = residual_block(x, source, w, f"gen.rb{i}", capture)
        if capture is not None:
            capture[f"gen.rb{i}.out"] = x.numpy().copy()
        x = upsample2x(x)
        i += 1
    return conv2d_same(leaky_relu(x, LEAK), w["gen.head.kernel"], w["gen.head.bias"])


def spade_call(source, weights, variant: str = "gaugan", eps=None, dtype=torch.float64,
               return_latent: bool = False, capture=None):
    """The reference's generator(call): source [B,S,S,2] -> [B,S,S,1].

    gaugan:        z = mean + exp(0.5*variance) * eps   (sampling.py:11-17; eps injected, shape [B, latent])
    gaugan_no_kl / cnn:  z = mean + variance              (model.py:265-267, 789-791)
    """
    w = {k: _t(v, dtype) for k, v in weights.items()}
    src = _t(source, dtype)
    with torch.no_grad():
        mean, variance = encoder(src, w, capture)
        if variant == "gaugan":
            if eps is None:
                raise ValueError("variant 'gaugan' needs the sampler noise eps [B, latent_dim]")
            z = mean + torch.exp(0.5 * variance) * _t(eps, dtype)
        elif variant in ("gaugan_no_kl", "cnn"):
            z = mean + variance
        else:
            raise ValueError(f"unknown SPADE variant {variant!r}")
        if capture is not None:
            capture["enc.mean"] = mean.numpy().copy()
            capture["enc.variance"] = variance.numpy().copy()
            capture["z"] = z.numpy().copy()
        out = generator(z, src, w, capture)
    if return_latent:
        return out.numpy(), z.numpy()
    return out.numpy()


def batch_norm_inference(x, w, prefix: str):
    """keras BatchNormalization with training=False: moving statistics, eps 1e-3."""
    return ((x - w[f"{prefix}.moving_mean"]) / torch.sqrt(w[f"{prefix}.moving_variance"] + BN_EPS)
            * w[f"{prefix}.gamma"] + w[f"{prefix}.beta"])


def pix2pix_call(source, weights, dtype=torch.float64):
    """Pix2Pix().generator(x, training=False), pix2pix.py:65-108 (dropout off, BN on moving stats)."""
    w = {k: _t(v, dtype) for k, v in weights.items()}
    x = _t(source, dtype)
    with torch.no_grad():
        skips = []
        i = 1
        while f"p2p.down{i}.kernel" in w:
            x = conv2d_same(x, w[f"p2p.down{i}.kernel"], None, stride=2)
            if f"p2p.down{i}.bn.gamma" in w:
                x = batch_norm_inference(x, w, f"p2p.down{i}.bn")
            x = leaky_relu(x, P2P_LEAK)
            skips.append(x)
            i += 1
        skips = list(reversed(skips[:-1]))
        i = 1
        while f"p2p.up{i}.kernel" in w:
            x = conv2d_transpose_same_s2(x, w[f"p2p.up{i}.kernel"])
            x = torch.relu(batch_norm_inference(x, w, f"p2p.up{i}.bn"))
            x = torch.cat([x, skips[i - 1]], dim=-1)
            i += 1
        x = torch.tanh(conv2d_transpose_same_s2(x, w["p2p.last.kernel"], w["p2p.last.bias"]))
    return x.numpy()
