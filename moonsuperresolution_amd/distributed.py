"""Multi-GPU sharding of the tiled path: one process per GPU, tiles partitioned by tile row.

The reference is single-process; its author notes the tile list "can be used to distribute the load"
(process_full_tiles.py:319-320).  Tiles are independent units (each re-computes its halo patches and its own
batches: process_full_tiles.py:449-474), so the exact mode needs NO collective on the data path — batch
composition, and with it SPADE's batch statistics, stay bit-for-bit the reference's.  The only exchange is the
gather of finished rows at the end, done with torch.distributed (backend "nccl" = RCCL over xGMI
on GPUs, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def tile_rows(tiles: Sequence[Tuple[int, int]]) -> List[int]:
    return sorted({yy for _, yy in tiles})


def rows_of_rank(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """(first, count) of the contiguous block of tile rows rank owns; sizes differ by at most one."""
    base, extra = divmod(n_rows, world)
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


def shard_tile_rows(tiles: Sequence[Tuple[int, int]], rank: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous blocks of tile rows per rank, sizes differing by at most one (15 rows on 8 ranks: 2,2,2,2,2,2,2,1)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    rows = tile_rows(tiles)
    start, count = rows_of_rank(len(rows), rank, world)
    mine = set(rows[start:start + count])
    return [t for t in tiles if t[1] in mine]


def all_gather_rows(local, n_rows: int, T: int, world: int):
    """All-gather the finished rows of one product over the process group (RCCL on GPUs, gloo on CPU tensors).

    local: this rank's [count * T, width] tensor (count = rows_of_rank(n_rows, rank, world)[1]).  Ranks own different
    numbers of tile rows, so every rank contributes a block padded to the largest count; ONE collective writes all
    blocks into one [world * max, width] buffer, which is then compacted in place (ranks with the larger count come
    first, so only blocks after the first short rank move, towards the front) and returned as [n_rows * T, width].
    Peak memory = that buffer + the padded block — not world copies of it."""
    import torch
    import torch.distributed as dist
    base, extra = divmod(n_rows, world)
    max_rows = (base + (1 if extra else 0)) * T
    width = local.shape[1]
    if local.shape[0] == max_rows:
        block = local.contiguous()
    else:
        block = torch.zeros((max_rows, width), dtype=local.dtype, device=local.device)
        block[:local.shape[0]] = local
    out = torch.empty((world * max_rows, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, block)
    if extra:   # ranks >= extra hold base rows: close the T-row gap behind each of them
        dst = (extra * (base + 1) + base) * T
        for r in range(extra + 1, world):
            src = r * max_rows
            out[dst:dst + base * T] = out[src:src + base * T].clone()
            dst += base * T
    return out[:n_rows * T]


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
        mean, std, good = (all_gather_rows(t, len(rows), T, world) for t in (mean, std, good))
    elif world > 1:
        full = [torch.zeros((len(rows) * T, width), dtype=t.dtype, device=dev) for t in (mean, std, good)]
        if my_rows:
            r0 = rows.index(my_rows[0]) * T
            for f, t in zip(full, (mean, std, good)):
                f[r0:r0 + t.shape[0]] = t
        mean, std, good = full
    return tuple(t[:h, :w].cpu().numpy() for t in (mean, std, good))


# ----------------------------------------------------------------------------------------------------------------
# Patch-row-sharded ("halo") mode: the exchange step (moonsuperresolution_amd/halo.py holds the GPU side)
# ----------------------------------------------------------------------------------------------------------------
def exchange_halo(send_down, send_up, recv_down_shape, recv_up_shape, rank: int, world: int):
    """Neighbour exchange of the boundary-zone accumulators: every rank sends `send_down` to rank - 1 and `send_up` to
    rank + 1 and receives tensors of the given shapes from them (None at the ends of the chain).  Point-to-point
    send / recv over the process group — RCCL on device tensors (xGMI is point-to-point: neighbour traffic needs no
    ring collective), gloo on host tensors in the CPU tests.  Returns (from_down, from_up)."""
    import torch
    import torch.distributed as dist
    ops, from_down, from_up = [], None, None
    like = send_down if send_down is not None else send_up
    if rank > 0:
        from_down = torch.empty(recv_down_shape, dtype=like.dtype, device=like.device)
        ops.append(dist.P2POp(dist.isend, send_down.contiguous(), rank - 1))
        ops.append(dist.P2POp(dist.irecv, from_down, rank - 1))
    if rank < world - 1:
        from_up = torch.empty(recv_up_shape, dtype=like.dtype, device=like.device)
        ops.append(dist.P2POp(dist.isend, send_up.contiguous(), rank + 1))
        ops.append(dist.P2POp(dist.irecv, from_up, rank + 1))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return from_down, from_up


def exchange_halo_start(send_down, send_up, recv_down_shape, recv_up_shape, rank: int, world: int):
    """exchange_halo split in two: the non-blocking sends / receives are started now, the returned ``wait()`` completes them
    and returns (from_down, from_up).  What the caller launches in between (the finalisation of its interior rows) runs
    beside the transfers."""
    import torch
    import torch.distributed as dist
    ops, from_down, from_up = [], None, None
    like = send_down if send_down is not None else send_up
    keep = []
    if rank > 0:
        from_down = torch.empty(recv_down_shape, dtype=like.dtype, device=like.device)
        keep.append(send_down.contiguous())
        ops.append(dist.P2POp(dist.isend, keep[-1], rank - 1))
        ops.append(dist.P2POp(dist.irecv, from_down, rank - 1))
    if rank < world - 1:
        from_up = torch.empty(recv_up_shape, dtype=like.dtype, device=like.device)
        keep.append(send_up.contiguous())
        ops.append(dist.P2POp(dist.isend, keep[-1], rank + 1))
        ops.append(dist.P2POp(dist.irecv, from_up, rank + 1))
    reqs = dist.batch_isend_irecv(ops) if ops else []

    def wait():
        for req in reqs:
            req.wait()
        keep.clear()
        return from_down, from_up
    return wait


def all_gather_var_rows(local, counts: Sequence[int]):
    """All-gather of row slabs whose row counts differ by rank (halo mode: ownership boundaries follow patch rows, not
    tiles): pad to the largest count, one all_gather_into_tensor, trim.  counts[r] = rows of rank r."""
    import torch
    import torch.distributed as dist
    world = len(counts)
    mx = max(counts)
    block = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    block[:local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, block)
    return torch.cat([out[r * mx:r * mx + counts[r]] for r in range(world)], dim=0)


def halo_zone_rows(ys: Sequence[int], image_size: int, world: int):
    """Geometry of the halo mode for patch-row origins `ys` (sorted, padded-canvas coordinates) split over `world`
    ranks: per rank (g0, g1) = its block of patch rows, (touch_lo, touch_hi) = the canvas rows its own patches reach
    (purged by S // 16 per side, like rebuildTile), (own_lo, own_hi) = the canvas rows it finalises.  Ownership passes
    from rank r - 1 to rank r at the centre of rank r's first patch row."""
    S, p = image_size, image_size // 16
    n = len(ys)
    out = []
    for r in range(world):
        g0, g1 = rows_of_rank(n, r, world)
        g1 = g0 + g1
        if g1 <= g0:
            raise ValueError("halo mode: more ranks than patch rows")
        out.append(dict(g0=g0, g1=g1, touch_lo=ys[g0] + p, touch_hi=ys[g1 - 1] + S - p))
    for r in range(world):
        out[r]["own_lo"] = 0 if r == 0 else ys[out[r]["g0"]] + S // 2
        out[r]["own_hi"] = None if r == world - 1 else ys[out[r + 1]["g0"]] + S // 2
    for r in range(world):       # a pixel may combine at most two ranks: every block must span a whole overlap
        if r + 2 < world and out[r]["touch_hi"] > out[r + 2]["touch_lo"]:
            raise ValueError("halo mode: too few patch rows per rank (a pixel would need three ranks)")
    return out
