"""Patch-row-sharded ("halo") mode of the tiled path — BASELINE.json north_star's "RCCL exchange of overlap halos".

NOT the reference's algorithm: process_full_tiles.py re-generates, for every 1024 x 1024 tile, the S - s wide halo of
patches around it (:449-454), so on a large raster every patch position is generated ~2x (1.995x at S = 512 / s = 64,
SURVEY.md section 5).  Here every position is generated ONCE: ranks own contiguous blocks of patch ROWS, each rank
accumulates the weighted incremental mean / variance of its own patches (msr_stitch_partial: rebuildTile stopped before
its finalisation), and the pixels near a block boundary — reached by patches of two ranks — combine the two ranks'
accumulators pairwise (msr_halo_merge) after a neighbour exchange (distributed.exchange_halo: RCCL send / recv).

Deviations from the reference (documented, inherent; oracle/tiler_ref.py::process_map_halo restates the mode):
  1. batches are cut from a rank's patch rows, not per tile, so SPADE's batch statistics (spade.py:21) see other batch
     mates: generator outputs differ beyond rounding (not for the identity model);
  2. the variance is textbook West, because the reference's aliased update (:400-402) has no pairwise combine;
  3. boundary-zone pixels are combined pairwise instead of sequentially: float32 rounding differs (~1e-7 relative).
The exact, reference-identical multi-GPU mode is the tile-row sharding of distributed.process_map_sharded.

Memory: a rank keeps the predictions of all its patches until it has stitched them (S * S * 4 bytes each: 34 k patches
= 34 GB per rank for the 15000 x 70000 raster on 8 ranks at S = 512 — sized for 288 GB of HBM).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .distributed import exchange_halo, halo_zone_rows
from .tiler import DEMSuperResolution


class HaloShardedSuperResolution(DEMSuperResolution):
    """DEMSuperResolution with the halo mode added: ``processMapHalo(rank, world)``."""

    def patchGrid(self) -> Tuple[list, list]:
        """Unique patch origins the reference's tiles touch (padded-canvas coordinates): (ys, xs), sorted."""
        S, s, T = self.image_size, self.stride, self.tile_size
        if S % s or T % s:
            # msr_stitch_partial bins patches by (origin - block origin) / stride: origins that are not multiples of the
            # stride relative to every T x T block would be dropped silently (and the union grid would be irregular)
            raise ValueError(f"halo mode needs a stride that divides image_size and tile_size (got stride {s}, "
                             f"image_size {S}, tile_size {T}); use the tile mode (processMap) for other strides")
        span = T + S - s
        ys = sorted({y for _, py in self.generateTileList() for y in range(py, py + span, s)})
        xs = sorted({x for px, _ in self.generateTileList() for x in range(px, px + span, s)})
        return ys, xs

    # ------------------------------------------------------------------------------------------------------------
    def _generate_rows(self, ys: Sequence[int], xs: Sequence[int]):
        """Validity, normalisation statistics, device-side batch assembly and the generator calls for every patch at
        rows `ys`: returns (preds [ncall * B, S, S], keys [cap, 2] canvas origins, dmm [cap, 2], valid count)."""
        S, B = self.image_size, self.batch_size
        lib, h, dev = self._lib, self._h, self.device
        rows, cols = self.dem_padded_shape
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            stream = cur.cuda_stream
            xt = torch.tensor(xs, dtype=torch.int32, device=dev)
            yt = torch.tensor(ys, dtype=torch.int32, device=dev)
            ox = xt.repeat(len(ys))
            oy = yt.repeat_interleave(len(xs))
            n = int(ox.numel())
            cap = max(B, (n + B - 1) // B * B)
            valid = torch.empty(n, dtype=torch.uint8, device=dev)
            minmax = torch.empty((n, 4), dtype=torch.float32, device=dev)
            sx = torch.empty(cap, dtype=torch.int32, device=dev)
            sy = torch.empty(cap, dtype=torch.int32, device=dev)
            mm_sel = torch.empty((cap, 4), dtype=torch.float32, device=dev)
            keys = torch.empty((cap, 2), dtype=torch.int32, device=dev)
            dmm = torch.empty((cap, 2), dtype=torch.float32, device=dev)
            meta = torch.empty(2, dtype=torch.int32, device=dev)
            rc = lib.msr_patch_stats(h, self.img_padded.data_ptr(), self.dem_padded.data_ptr(), rows, cols, ox.data_ptr(),
                                     oy.data_ptr(), n, self.no_value, valid.data_ptr(), minmax.data_ptr(), stream)
            _lib.raise_for(lib, h, rc, "msr_patch_stats")
            rc = lib.msr_compact_patches(h, valid.data_ptr(), ox.data_ptr(), oy.data_ptr(), minmax.data_ptr(), n, 0, 0, B,
                                         cap, sx.data_ptr(), sy.data_ptr(), mm_sel.data_ptr(), keys.data_ptr(),
                                         dmm.data_ptr(), meta.data_ptr(), stream)
            _lib.raise_for(lib, h, rc, "msr_compact_patches")
            nv, ncall = (int(v) for v in meta.cpu().tolist())
            preds = torch.empty((max(ncall * B, 1), S, S), dtype=torch.float32, device=dev)
            if self._gen is not None:
                if self._gens is None:
                    self._make_pipeline()
                batches = [torch.empty((B, S, S, 2), dtype=torch.float32, device=dev) for _ in self._gens]
                for ps in self._pstreams:
                    ps.wait_stream(cur)
                for c in range(ncall):
                    k = c % len(self._gens)
                    with torch.cuda.stream(self._pstreams[k]):
                        rc = lib.msr_extract_patches(h, self.img_padded.data_ptr(), self.dem_padded.data_ptr(), rows, cols,
                                                     sx[c * B:].data_ptr(), sy[c * B:].data_ptr(),
                                                     mm_sel[c * B:].data_ptr(), B, batches[k].data_ptr(), self._stream())
                        _lib.raise_for(lib, h, rc, "msr_extract_patches")
                        self._gens[k].forward_device(batches[k], out=preds[c * B:(c + 1) * B].unsqueeze(-1))
                for ps in self._pstreams:
                    cur.wait_stream(ps)
                    for t in batches + [preds, sx, sy, mm_sel]:
                        t.record_stream(ps)
            else:
                batch = torch.empty((B, S, S, 2), dtype=torch.float32, device=dev)
                for c in range(ncall):
                    rc = lib.msr_extract_patches(h, self.img_padded.data_ptr(), self.dem_padded.data_ptr(), rows, cols,
                                                 sx[c * B:].data_ptr(), sy[c * B:].data_ptr(), mm_sel[c * B:].data_ptr(), B,
                                                 batch.data_ptr(), stream)
                    _lib.raise_for(lib, h, rc, "msr_extract_patches")
                    out = np.array(self.model(batch.cpu().numpy(), training=False))[:, :, :, -1]
                    preds[c * B:(c + 1) * B] = torch.from_numpy(np.ascontiguousarray(out, dtype=np.float32)).to(dev)
            self.last_counts_halo = (nv, ncall)
            return preds, keys, dmm, nv

    def _accumulate(self, preds, keys, dmm, nv: int, row_lo: int, row_hi: int):
        """Accumulators (w_sum, mean, S) of the given patches for canvas rows [row_lo, row_hi), all canvas columns:
        [3, row_hi - row_lo, Wp] float32.  Stitched block by block (T x T) with the gather kernel of the tile mode."""
        T, S, s = self.tile_size, self.image_size, self.stride
        halo = S - s
        lib, h, dev = self._lib, self._h, self.device
        hp, wp = self.dem_padded_shape
        y_b0 = (row_lo // T) * T
        n_by = (row_hi - y_b0 + T - 1) // T
        n_bx = (wp + T - 1) // T
        with torch.cuda.device(dev):
            acc = torch.zeros((3, n_by * T, n_bx * T), dtype=torch.float32, device=dev)
            blk = torch.empty((3, T, T), dtype=torch.float32, device=dev)
            for by in range(n_by):
                for bx in range(n_bx):
                    y0, x0 = y_b0 + by * T, bx * T
                    # patch keys relative to the block's grid origin (x0 - halo, y0 - halo): accumulator coordinate =
                    # block pixel + halo, exactly the tile mode's geometry
                    rel = keys[:max(nv, 1)] - torch.tensor([x0 - halo, y0 - halo], dtype=torch.int32, device=dev)
                    rc = lib.msr_stitch_partial(h, preds.data_ptr(), rel.data_ptr(), dmm.data_ptr(), nv, T, s,
                                                blk[0].data_ptr(), blk[1].data_ptr(), blk[2].data_ptr(), self._stream())
                    _lib.raise_for(lib, h, rc, "msr_stitch_partial")
                    acc[:, by * T:(by + 1) * T, bx * T:(bx + 1) * T] = blk
            return acc[:, row_lo - y_b0:row_hi - y_b0, :wp].contiguous()

    def _finalize(self, a, b=None):
        """msr_halo_merge over [3, rows, W] accumulators -> (mean, std, good)."""
        rows, w = a.shape[1], a.shape[2]
        with torch.cuda.device(self.device):
            mean = torch.empty((rows, w), dtype=torch.float32, device=self.device)
            std = torch.empty_like(mean)
            good = torch.empty((rows, w), dtype=torch.uint8, device=self.device)
            if rows * w:
                a = a.contiguous()
                bp = [None, None, None]
                if b is not None:
                    b = b.contiguous()
                    bp = [b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr()]
                rc = self._lib.msr_halo_merge(self._h, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), bp[0], bp[1],
                                              bp[2], rows * w, self.no_value, mean.data_ptr(), std.data_ptr(),
                                              good.data_ptr(), self._stream())
                _lib.raise_for(self._lib, self._h, rc, "msr_halo_merge")
        return mean, std, good

    # ------------------------------------------------------------------------------------------------------------
    def haloAccumulate(self, rank: int = 0, world: int = 1):
        """Phase 1 (no communication): generate this rank's patch rows and accumulate them.  Returns the state
        ``haloFinish`` takes: the accumulators of the canvas rows the rank's patches reach, what it must send to its
        neighbours and the shapes of what it receives."""
        if self.dem_padded is None or self.dem is not None:
            self.padInputs()
        S = self.image_size
        hp, wp = self.dem_padded_shape
        ys, xs = self.patchGrid()
        zones = halo_zone_rows(ys, S, world)
        z = zones[rank]
        lo, hi = z["touch_lo"], z["touch_hi"]
        own_lo = z["own_lo"]
        own_hi = hp if z["own_hi"] is None else z["own_hi"]
        if (rank > 0 and own_lo < lo) or (rank < world - 1 and own_hi > hi):
            raise ValueError("halo mode needs overlapping patches (stride <= S/2 - S/16)")
        preds, keys, dmm, nv = self._generate_rows(ys[z["g0"]:z["g1"]], xs)
        acc = self._accumulate(preds, keys, dmm, nv, lo, hi)
        del preds
        st = dict(rank=rank, world=world, acc=acc, lo=lo, hi=hi, own_lo=own_lo, own_hi=own_hi, wp=wp,
                  send_down=acc[:, :own_lo - lo] if rank > 0 else None,                     # rows [touch_lo, own_lo)
                  send_up=acc[:, own_hi - lo:] if rank < world - 1 else None,                # rows [own_hi, touch_hi)
                  down_rows=max(0, zones[rank - 1]["touch_hi"] - own_lo) if rank > 0 else 0,
                  up_rows=max(0, own_hi - zones[rank + 1]["touch_lo"]) if rank < world - 1 else 0)
        return st

    def haloFinish(self, st, from_down=None, from_up=None):
        """Phase 2: combine the boundary zones with the neighbours' accumulators (from_down = rank - 1's rows
        [own_lo, own_lo + down_rows), its patches come first; from_up = rank + 1's rows [own_hi - up_rows, own_hi)) and
        finalise.  Returns ((mean, std, good) for canvas rows [own_lo, own_hi), (own_lo, own_hi))."""
        acc, lo, hi, own_lo, own_hi, wp = st["acc"], st["lo"], st["hi"], st["own_lo"], st["own_hi"], st["wp"]
        down_rows = st["down_rows"] if from_down is not None else 0
        up_rows = st["up_rows"] if from_up is not None else 0
        with torch.cuda.device(self.device):
            mean = torch.full((own_hi - own_lo, wp), self.no_value, dtype=torch.float32, device=self.device)
            std = torch.full_like(mean, self.no_value)
            good = torch.zeros((own_hi - own_lo, wp), dtype=torch.uint8, device=self.device)

            def put(r0, r1, a, b=None):
                if r1 > r0:
                    m, s_, g = self._finalize(a, b)
                    mean[r0 - own_lo:r1 - own_lo], std[r0 - own_lo:r1 - own_lo], good[r0 - own_lo:r1 - own_lo] = m, s_, g

            d_end, u_beg = own_lo + down_rows, own_hi - up_rows
            if down_rows:                                        # zone shared with the rank below
                put(own_lo, d_end, from_down, acc[:, own_lo - lo:d_end - lo])
            m0, m1 = max(lo, d_end), min(hi, u_beg)              # rows only my patches reach
            put(m0, m1, acc[:, m0 - lo:m1 - lo])
            if up_rows:                                          # zone shared with the rank above
                put(u_beg, own_hi, acc[:, u_beg - lo:own_hi - lo], from_up)
        return (mean, std, good), (own_lo, own_hi)

    def processMapHalo(self, img: Optional[np.ndarray] = None, dem: Optional[np.ndarray] = None, rank: int = 0,
                       world: int = 1, exchange: Optional[Callable] = None):
        """This rank's share of the map in halo mode: accumulate, exchange the boundary zones with the neighbours
        (``exchange`` defaults to distributed.exchange_halo: torch.distributed send / recv, RCCL on the GPUs), finish."""
        if img is not None:
            self.setImages(img, dem)
        st = self.haloAccumulate(rank, world)
        from_down = from_up = None
        if world > 1:
            torch.cuda.current_stream(self.device).synchronize()
            wp = st["wp"]
            from_down, from_up = (exchange or exchange_halo)(st["send_down"], st["send_up"], (3, st["down_rows"], wp),
                                                             (3, st["up_rows"], wp), rank, world)
        return self.haloFinish(st, from_down, from_up)

    def cropHalo(self, slabs: Sequence[Tuple[Tuple[torch.Tensor, torch.Tensor, torch.Tensor], Tuple[int, int]]]):
        """Assemble per-rank slabs (in rank order) into the final rasters cropped to the input extent: final pixel
        (y, x) is canvas pixel (y + S - s, x + S - s)."""
        halo = self.image_size - self.stride
        h, w = self.dem_shape
        parts = [torch.cat([sl[0][k] for sl in slabs], dim=0) for k in range(3)]
        return tuple(p[halo:halo + h, halo:halo + w].cpu().numpy() for p in parts)
