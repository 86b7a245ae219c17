"""Weight naming, shapes and a framework-independent seeded initialiser.

The reference keeps its weights in Keras SavedModel directories
(spade/models/model.py:569-610) and ships none, so random-init weights with the
Keras default distributions stand in for them:

* Conv2D / Dense kernels: glorot_uniform, bias zeros (Keras defaults used by
  spade/models/spade.py:9-11, blocks.py:19-26, networks.py:32-33,41,56).
* encoder convs: GlorotNormal (blocks.py:59).
* InstanceNormalization: gamma 1, beta 0 (tfa default, blocks.py:63).
* pix2pix convs: N(0, 0.02) (pix2pix.py:66,77,90); BatchNormalization
  gamma 1, beta 0, moving_mean 0, moving_variance 1 (Keras defaults).

Layouts are the reference's own: conv kernels HWIO ``[kh, kw, Cin, Cout]``,
transposed-conv kernels ``[kh, kw, Cout, Cin]``, dense ``[in, out]``.

Every tensor is drawn from ``numpy.random.default_rng([seed, crc32(name)])`` so
the oracle and the HIP side regenerate bit-identical arrays on any machine
without shipping 0.5-0.7 GB fixtures.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

SPADE_VARIANTS = ("gaugan", "gaugan_no_kl", "cnn")
VARIANTS = SPADE_VARIANTS + ("pix2pix",)

# generator ResidualBlock filters (networks.py:43-53)
GEN_FILTERS = (1024, 1024, 1024, 512, 256, 128)
# encoder downsample channels with encoder_downsample_factor=64 (networks.py:16-30, model.py:373-379)
ENC_CHANNELS = (64, 128, 256, 512, 512)
SPADE_HIDDEN = 128  # spade.py:9
# pix2pix stacks (pix2pix.py:10-28)
P2P_DOWN = (64, 128, 256, 512, 512, 512, 512, 512)
P2P_UP = (512, 512, 512, 512, 256, 128, 64)


def spade_shapes(image_size: int, latent_dim: int = 256) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape for the encoder + generator of GauGAN/GauGAN_no_KL/CNNSpade."""
    if image_size % 64 or image_size < 64:
        raise ValueError("image_size must be a positive multiple of 64 (networks.py:40: sw = S // 2**6)")
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    cin = 2
    for i, c in enumerate(ENC_CHANNELS, start=1):
        shapes[f"enc.ds{i}.kernel"] = (3, 3, cin, c)
        if i > 1:
            shapes[f"enc.ds{i}.in.gamma"] = (c,)
            shapes[f"enc.ds{i}.in.beta"] = (c,)
        cin = c
    flat = (image_size // 32) ** 2 * ENC_CHANNELS[-1]
    for head in ("mean", "variance"):
        shapes[f"enc.{head}.kernel"] = (flat, latent_dim)
        shapes[f"enc.{head}.bias"] = (latent_dim,)
    sw = image_size // 64
    shapes["gen.dense.kernel"] = (latent_dim, sw * sw * 1024)
    shapes["gen.dense.bias"] = (sw * sw * 1024,)
    cin = 1024
    for i, f in enumerate(GEN_FILTERS, start=1):
        learned = f != cin
        spades = ((1, cin), (2, f)) + (((3, cin),) if learned else ())
        for j, c in spades:
            p = f"gen.rb{i}.spade_{j}"
            shapes[f"{p}.conv.kernel"] = (3, 3, 2, SPADE_HIDDEN)
            shapes[f"{p}.conv.bias"] = (SPADE_HIDDEN,)
            shapes[f"{p}.conv_gamma.kernel"] = (3, 3, SPADE_HIDDEN, c)
            shapes[f"{p}.conv_gamma.bias"] = (c,)
            shapes[f"{p}.conv_beta.kernel"] = (3, 3, SPADE_HIDDEN, c)
            shapes[f"{p}.conv_beta.bias"] = (c,)
        shapes[f"gen.rb{i}.conv_1.kernel"] = (3, 3, cin, f)
        shapes[f"gen.rb{i}.conv_1.bias"] = (f,)
        shapes[f"gen.rb{i}.conv_2.kernel"] = (3, 3, f, f)
        shapes[f"gen.rb{i}.conv_2.bias"] = (f,)
        if learned:
            shapes[f"gen.rb{i}.conv_3.kernel"] = (3, 3, cin, f)
            shapes[f"gen.rb{i}.conv_3.bias"] = (f,)
        cin = f
    shapes["gen.head.kernel"] = (4, 4, GEN_FILTERS[-1], 1)
    shapes["gen.head.bias"] = (1,)
    return shapes


def pix2pix_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape for Pix2Pix().generator (pix2pix.py:88-108)."""
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    cin = 2
    for i, c in enumerate(P2P_DOWN, start=1):
        shapes[f"p2p.down{i}.kernel"] = (4, 4, cin, c)
        if i > 1:
            for n in ("gamma", "beta", "moving_mean", "moving_variance"):
                shapes[f"p2p.down{i}.bn.{n}"] = (c,)
        cin = c
    skips = list(reversed(P2P_DOWN[:-1]))
    for i, c in enumerate(P2P_UP, start=1):
        shapes[f"p2p.up{i}.kernel"] = (4, 4, c, cin)  # Conv2DTranspose: [kh, kw, Cout, Cin]
        for n in ("gamma", "beta", "moving_mean", "moving_variance"):
            shapes[f"p2p.up{i}.bn.{n}"] = (c,)
        cin = c + skips[i - 1]
    shapes["p2p.last.kernel"] = (4, 4, 1, cin)
    shapes["p2p.last.bias"] = (1,)
    return shapes


def weight_shapes(variant: str, image_size: int, latent_dim: int = 256):
    if variant in SPADE_VARIANTS:
        return spade_shapes(image_size, latent_dim)
    if variant == "pix2pix":
        return pix2pix_shapes()
    raise ValueError(f"unknown variant {variant!r}; expected one of {VARIANTS}")


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def _fans(shape: Tuple[int, ...], transposed: bool) -> Tuple[int, int]:
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = shape[0] * shape[1]
    cin, cout = (shape[3], shape[2]) if transposed else (shape[2], shape[3])
    return rf * cin, rf * cout


def init_tensor(name: str, shape: Tuple[int, ...], seed: int, bias_scale: float = 0.0) -> np.ndarray:
    """One tensor, float32, from the distribution the reference's layer would use."""
    rng = _rng(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("gamma", "moving_variance"):
        out = np.ones(shape, np.float32)
        if bias_scale:
            out += (bias_scale * rng.uniform(-1.0, 1.0, shape)).astype(np.float32)
        return out
    if leaf in ("bias", "beta", "moving_mean"):
        if bias_scale:
            return (bias_scale * rng.uniform(-1.0, 1.0, shape)).astype(np.float32)
        return np.zeros(shape, np.float32)
    if leaf != "kernel":
        raise ValueError(f"don't know how to initialise {name!r}")
    if name.startswith("p2p."):
        return (0.02 * rng.standard_normal(shape)).astype(np.float32)
    fan_in, fan_out = _fans(shape, transposed=False)
    if name.startswith("enc.ds"):
        # GlorotNormal = truncated normal (|z| <= 2) rescaled to stddev sqrt(2/(fi+fo))
        std = np.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
        z = np.clip(rng.standard_normal(shape), -2.0, 2.0)
        return (std * z).astype(np.float32)
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, shape).astype(np.float32)


def make_weights(variant: str, image_size: int, latent_dim: int = 256, seed: int = 1234,
                 bias_scale: float = 0.0) -> Dict[str, np.ndarray]:
    """Seeded random-init weights for ``variant`` (weights seed 1234).

    ``bias_scale`` > 0 perturbs every bias / beta / gamma / moving statistic away
    from the Keras defaults (0 / 1) so parity tests also exercise those terms.
    """
    return OrderedDict(
        (n, init_tensor(n, s, seed, bias_scale))
        for n, s in weight_shapes(variant, image_size, latent_dim).items()
    )


def make_latent_noise(batch_size: int, latent_dim: int = 256, seed: int = 7) -> np.ndarray:
    """The epsilon ~ N(0,1) of GaussianSampler (sampling.py:13-15), injected so runs are repeatable."""
    return np.random.default_rng(seed).standard_normal((batch_size, latent_dim)).astype(np.float32)


def synthetic_patches(batch: int, image_size: int, seed: int = 0) -> np.ndarray:
    """Smooth random (ortho, DEM) patches, min-max normalised to [-0.5, 0.5] per patch and channel.

    Follows the input convention of process_full_tiles.py:307-310 (channel 0 = ortho, 1 = DEM)
    and the synthetic-input recipe of SURVEY.md 8d.
    """
    S = image_size
    out = np.empty((batch, S, S, 2), np.float32)
    for b in range(batch):
        rng = np.random.default_rng([seed, b])
        g = max(S // 16, 2)
        coarse = rng.uniform(0.0, 1.0, (2, g + 1, g + 1))
        ys = np.linspace(0.0, g, S, endpoint=False)
        y0 = np.floor(ys).astype(int)
        fy = (ys - y0)[:, None]
        fx = (ys - y0)[None, :]
        for c in range(2):
            t = coarse[c]
            f = (t[y0][:, y0] * (1 - fy) * (1 - fx) + t[y0 + 1][:, y0] * fy * (1 - fx)
                 + t[y0][:, y0 + 1] * (1 - fy) * fx + t[y0 + 1][:, y0 + 1] * fy * fx)
            if c == 0:
                f = 0.7 * f + 0.3 * rng.uniform(0.0, 1.0, (S, S))
            f = f.astype(np.float32)
            out[b, :, :, c] = (f - f.min()) / (f.max() - f.min()) - np.float32(0.5)
    return out
