"""Multi-GPU sharding of the tiled path: one process per GPU, tiles partitioned by tile row.

The reference is single-process; its author notes the tile list "can be used to distribute the load"
(process_full_tiles.py:319-320).  Tiles are independent units (each re-computes its halo patches and its own
batches: process_full_tiles.py:449-474), so the exact mode needs NO collective on the data path — batch
composition, and with it SPADE's batch statistics, stay bit-for-bit the reference's.  The only exchange is the
gather of finished rows at the end, done with torch.distributed (backend "nccl" = RCCL over xGMI
on GPUs, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np


def tile_rows(tiles: Sequence[Tuple[int, int]]) -> List[int]:
    return sorted({yy for _, yy in tiles})


def shard_tile_rows(tiles: Sequence[Tuple[int, int]], rank: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous blocks of tile rows per rank, sizes differing by at most one (15 rows on 8 ranks: 2,2,2,2,2,2,2,1)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    rows = tile_rows(tiles)
    base, extra = divmod(len(rows), world)
    start = rank * base + min(rank, extra)
    mine = set(rows[start:start + base + (1 if rank < extra else 0)])
    return [t for t in tiles if t[1] in mine]


def process_map_sharded(shape: Tuple[int, int], tile_size: int, tiles: Sequence[Tuple[int, int]],
                        process_tile: Callable[[int, int], Tuple[np.ndarray, np.ndarray, np.ndarray]],
                        rank: int = 0, world: int = 1, gather: bool = True, device=None):
    """Run this rank's tile rows and (optionally) all-gather the finished rows into the full map on every rank.

    process_tile(xx, yy) -> (mean f32 [T,T], std f32 [T,T], good u8 [T,T]) (host arrays or torch tensors).
    Returns (mean, std, good) cropped to ``shape``; without ``gather`` rows of other ranks stay zero.
    """
    import torch
    h, w = shape
    T = tile_size
    rows = tile_rows(tiles)
    ncols = len({xx for xx, _ in tiles})
    width = ncols * T
    mine = shard_tile_rows(tiles, rank, world)
    my_rows = tile_rows(mine) if mine else []
    dev = device if device is not None else "cpu"
    mean = torch.zeros((len(my_rows) * T, width), dtype=torch.float32, device=dev)
    std = torch.zeros_like(mean)
    good = torch.zeros((len(my_rows) * T, width), dtype=torch.uint8, device=dev)
    for xx, yy in mine:
        m, s, g = process_tile(xx, yy)
        r0 = my_rows.index(yy) * T
        mean[r0:r0 + T, xx:xx + T] = torch.as_tensor(m).to(dev)
        std[r0:r0 + T, xx:xx + T] = torch.as_tensor(s).to(dev)
        good[r0:r0 + T, xx:xx + T] = torch.as_tensor(g).to(dev)
    if world > 1 and gather:
        import torch.distributed as dist
        # ranks own different numbers of rows: pad to the maximum, all_gather, trim
        base, extra = divmod(len(rows), world)
        max_rows = (base + (1 if extra else 0)) * T
        outs = []
        for t in (mean, std, good):
            pad = torch.zeros((max_rows, width), dtype=t.dtype, device=dev)
            pad[:t.shape[0]] = t
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad)
            outs.append(torch.cat([p[:(base + (1 if r < extra else 0)) * T] for r, p in enumerate(parts)], dim=0))
        mean, std, good = outs
    elif world > 1:
        full = [torch.zeros((len(rows) * T, width), dtype=t.dtype, device=dev) for t in (mean, std, good)]
        if my_rows:
            r0 = rows.index(my_rows[0]) * T
            for f, t in zip(full, (mean, std, good)):
                f[r0:r0 + t.shape[0]] = t
        mean, std, good = full
    return tuple(t[:h, :w].cpu().numpy() for t in (mean, std, good))
