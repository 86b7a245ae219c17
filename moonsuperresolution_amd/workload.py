"""Derived workload constants of the hot path (host logic; pinned against BASELINE.md section 2 in tests).

Counting rule: FLOPs = 2 * sum over convs (out_h*out_w*Cin*Cout*kh*kw) + 2 * sum over dense
(in*out); SAME-padded borders counted as full; normalisation / activation / resize not counted.
"""
from __future__ import annotations

from typing import Dict, Tuple

from .weights import ENC_CHANNELS, GEN_FILTERS, P2P_DOWN, P2P_UP, SPADE_HIDDEN, weight_shapes


def spade_flops_per_patch(image_size: int, latent_dim: int = 256) -> Dict[str, float]:
    """Algorithmic FLOPs of one GauGAN.call patch, split by the families BASELINE.md lists."""
    S = image_size
    f = dict(encoder=0.0, dense=0.0, mask_embed=0.0, spade_gamma_beta=0.0, resblock=0.0, head=0.0)
    cin, r = 2, S
    for c in ENC_CHANNELS:
        r //= 2
        f["encoder"] += 2.0 * r * r * cin * c * 9
        cin = c
    f["dense"] += 2.0 * (r * r * cin) * latent_dim * 2
    sw = S // 64
    f["dense"] += 2.0 * latent_dim * sw * sw * 1024
    cin = 1024
    for i, flt in enumerate(GEN_FILTERS):
        r = sw << i
        learned = flt != cin
        for c in (cin, flt) + ((cin,) if learned else ()):
            f["mask_embed"] += 2.0 * r * r * 2 * SPADE_HIDDEN * 9
            f["spade_gamma_beta"] += 2.0 * r * r * SPADE_HIDDEN * (2 * c) * 9
        f["resblock"] += 2.0 * r * r * 9 * (cin * flt + flt * flt + (cin * flt if learned else 0))
        cin = flt
    f["head"] = 2.0 * S * S * 16 * cin
    f["total"] = sum(f.values())
    return f


def pix2pix_flops_per_patch() -> float:
    """Pix2Pix().generator on a 256x256 patch; a stride-2 Conv2DTranspose does in_h*in_w*Cin*Cout*k*k MACs."""
    total, cin, r = 0.0, 2, 256
    for c in P2P_DOWN:
        r //= 2
        total += 2.0 * r * r * cin * c * 16
        cin = c
    skips = list(reversed(P2P_DOWN[:-1]))
    for i, c in enumerate(P2P_UP):
        total += 2.0 * r * r * cin * c * 16
        r *= 2
        cin = c + skips[i]
    total += 2.0 * r * r * cin * 1 * 16
    return total


def param_count(variant: str, image_size: int, latent_dim: int = 256) -> int:
    n = 0
    for name, shape in weight_shapes(variant, image_size, latent_dim).items():
        if name.endswith(("moving_mean", "moving_variance")):
            continue   # non-trainable statistics
        k = 1
        for d in shape:
            k *= d
        n += k
    return n


def raster_geometry(shape: Tuple[int, int], image_size: int, stride: int, tile_size: int = 1024) -> Dict[str, int]:
    """Canvas, tile and patch counts of process_full_tiles.py:246-267, 313-325, 453-454 (all-valid upper bound)."""
    h, w = shape
    halo = image_size - stride
    per_side = len(range(0, tile_size + halo, stride))
    tiles_y = len(range(0, h, tile_size))
    tiles_x = len(range(0, w, tile_size))
    return dict(canvas_rows=((h // 1024) + 1) * 1024 + 2 * halo, canvas_cols=((w // 1024) + 1) * 1024 + 2 * halo,
                tiles_y=tiles_y, tiles_x=tiles_x, tiles=tiles_x * tiles_y, patches_per_tile=per_side * per_side,
                patches=tiles_x * tiles_y * per_side * per_side)
