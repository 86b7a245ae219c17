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

Memory (round 3): a rank works through its patch rows in BANDS (``band_rows`` patch rows at a time, default sized to
``band_bytes`` = 4 GiB of predictions): a band is generated, accumulated into the canvas rows it reaches — in place, block by
block, with only the band's own keys (msr_stitch_accumulate: the running update resumes from what earlier bands left, so the
result does not depend on the band size, bit for bit) — and its predictions are freed.  What stays resident is the rank's
accumulator slab (3 float32 images of the canvas rows its patches reach: 2.1 GB for a 1/8 share of the 15000 x 70000 raster)
plus one band; round 2 kept all predictions of the rank (34 GB at that size).
Exchange: the two boundary-zone slabs go to the neighbours as non-blocking send / recv while the interior rows (reached by this
rank's patches only) are finalised; the zones are finalised when the neighbours' slabs have arrived.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .distributed import halo_zone_rows
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

    def _accumulate_band(self, acc, y_b0: int, preds, keys, dmm, nv: int, band_lo: int, band_hi: int) -> None:
        """Add the patches of one band (canvas rows [band_lo, band_hi) are the rows they reach) to the accumulator slab
        ``acc`` [3, n_by * T, n_bx * T] whose row 0 is canvas row ``y_b0`` (a multiple of T): every T x T block the band
        touches is continued in place from its current content, with the band's keys only."""
        T, S, s = self.tile_size, self.image_size, self.stride
        halo = S - s
        lib, h, dev = self._lib, self._h, self.device
        pitch = acc.shape[2]
        by0, by1 = (band_lo - y_b0) // T, (band_hi - 1 - y_b0) // T
        with torch.cuda.device(dev):
            for by in range(max(by0, 0), min(by1, acc.shape[1] // T - 1) + 1):
                for bx in range(pitch // T):
                    y0, x0 = y_b0 + by * T, bx * T
                    # patch keys relative to the block's grid origin (x0 - halo, y0 - halo): accumulator coordinate =
                    # block pixel + halo, exactly the tile mode's geometry
                    rel = keys[:max(nv, 1)] - torch.tensor([x0 - halo, y0 - halo], dtype=torch.int32, device=dev)
                    win = acc[:, by * T:(by + 1) * T, bx * T:(bx + 1) * T]
                    rc = lib.msr_stitch_accumulate(h, preds.data_ptr(), rel.data_ptr(), dmm.data_ptr(), nv, T, s,
                                                   win[0].data_ptr(), win[1].data_ptr(), win[2].data_ptr(), pitch, 1,
                                                   self._stream())
                    _lib.raise_for(lib, h, rc, "msr_stitch_accumulate")

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
    def haloAccumulate(self, rank: int = 0, world: int = 1, band_rows: Optional[int] = None, band_bytes: int = 4 << 30,
                       max_rows: int = 0):
        """Phase 1 (no communication): generate this rank's patch rows band by band and accumulate them.  Returns the
        state ``haloFinish`` takes: the accumulators of the canvas rows the rank's patches reach, what it must send to its
        neighbours and the shapes of what it receives.  ``band_rows`` patch rows per band (default: as many as fit
        ``band_bytes`` of predictions); ``max_rows`` > 0 stops after that many patch rows (benchmarks of a share of a
        large raster: the accumulators of the rows not reached stay empty)."""
        if self.dem_padded is None or self.dem is not None:
            self.padInputs()
        S, T = self.image_size, self.tile_size
        p = S // 16
        hp, wp = self.dem_padded_shape
        ys, xs = self.patchGrid()
        zones = halo_zone_rows(ys, S, world)
        z = zones[rank]
        lo, hi = z["touch_lo"], z["touch_hi"]
        own_lo = z["own_lo"]
        own_hi = hp if z["own_hi"] is None else z["own_hi"]
        if (rank > 0 and own_lo < lo) or (rank < world - 1 and own_hi > hi):
            raise ValueError("halo mode needs overlapping patches (stride <= S/2 - S/16)")
        my_ys = ys[z["g0"]:z["g1"]]
        if band_rows is None:
            band_rows = max(1, int(band_bytes // max(1, len(xs) * S * S * 4)))
        y_b0 = (lo // T) * T
        n_by = (hi - y_b0 + T - 1) // T
        n_bx = (wp + T - 1) // T
        nv_tot = nc_tot = 0
        with torch.cuda.device(self.device):
            slab = torch.zeros((3, n_by * T, n_bx * T), dtype=torch.float32, device=self.device)
            done = 0
            for b0 in range(0, len(my_ys), band_rows):
                band = my_ys[b0:b0 + band_rows]
                if max_rows and done + len(band) > max_rows:
                    band = band[:max_rows - done]
                if not band:
                    break
                preds, keys, dmm, nv = self._generate_rows(band, xs)
                self._accumulate_band(slab, y_b0, preds, keys, dmm, nv, band[0] + p, band[-1] + S - p)
                nv_tot += self.last_counts_halo[0]
                nc_tot += self.last_counts_halo[1]
                done += len(band)
                del preds, keys, dmm                     # the band's predictions are not needed again
            acc = slab[:, lo - y_b0:hi - y_b0, :wp]
        self.last_counts_halo = (nv_tot, nc_tot)
        self.last_band_rows = band_rows
        st = dict(rank=rank, world=world, acc=acc, lo=lo, hi=hi, own_lo=own_lo, own_hi=own_hi, wp=wp,
                  send_down=acc[:, :own_lo - lo] if rank > 0 else None,                     # rows [touch_lo, own_lo)
                  send_up=acc[:, own_hi - lo:] if rank < world - 1 else None,                # rows [own_hi, touch_hi)
                  down_rows=max(0, zones[rank - 1]["touch_hi"] - own_lo) if rank > 0 else 0,
                  up_rows=max(0, own_hi - zones[rank + 1]["touch_lo"]) if rank < world - 1 else 0)
        return st

    def haloFinish(self, st, from_down=None, from_up=None, *, exchange: Optional[Callable] = None):
        """Phase 2: combine the boundary zones with the neighbours' accumulators (from_down = rank - 1's rows
        [own_lo, own_lo + down_rows), its patches come first; from_up = rank + 1's rows [own_hi - up_rows, own_hi)) and
        finalise.  With ``exchange`` (a callable returning (from_down, from_up), e.g. the ``wait`` of
        distributed.exchange_halo_start) the interior rows are finalised FIRST and the exchange is waited for afterwards:
        the transfer runs beside the interior kernels.
        Returns ((mean, std, good) for canvas rows [own_lo, own_hi), (own_lo, own_hi))."""
        acc, lo, hi, own_lo, own_hi, wp = st["acc"], st["lo"], st["hi"], st["own_lo"], st["own_hi"], st["wp"]
        with torch.cuda.device(self.device):
            mean = torch.full((own_hi - own_lo, wp), self.no_value, dtype=torch.float32, device=self.device)
            std = torch.full_like(mean, self.no_value)
            good = torch.zeros((own_hi - own_lo, wp), dtype=torch.uint8, device=self.device)

            def put(r0, r1, a, b=None):
                # in slabs of T rows: the accumulator is a view with the slab's pitch, so each piece is compacted for
                # the merge kernel — T rows at a time instead of a second copy of the whole interior
                for c0 in range(r0, r1, self.tile_size):
                    c1 = min(r1, c0 + self.tile_size)
                    m, s_, g = self._finalize(a[:, c0 - r0:c1 - r0], None if b is None else b[:, c0 - r0:c1 - r0])
                    mean[c0 - own_lo:c1 - own_lo], std[c0 - own_lo:c1 - own_lo], good[c0 - own_lo:c1 - own_lo] = m, s_, g

            # rows only my patches reach: independent of the neighbours (the zone row counts are known from the geometry)
            d_end = own_lo + (st["down_rows"] if st["rank"] > 0 else 0)
            u_beg = own_hi - (st["up_rows"] if st["rank"] < st["world"] - 1 else 0)
            if exchange is None and from_down is None:
                d_end = own_lo
            if exchange is None and from_up is None:
                u_beg = own_hi
            m0, m1 = max(lo, d_end), min(hi, u_beg)
            put(m0, m1, acc[:, m0 - lo:m1 - lo])
            if exchange is not None:
                from_down, from_up = exchange()
            if d_end > own_lo:                                   # zone shared with the rank below
                put(own_lo, d_end, from_down, acc[:, own_lo - lo:d_end - lo])
            if own_hi > u_beg:                                   # zone shared with the rank above
                put(u_beg, own_hi, acc[:, u_beg - lo:own_hi - lo], from_up)
        return (mean, std, good), (own_lo, own_hi)

    def processMapHalo(self, img: Optional[np.ndarray] = None, dem: Optional[np.ndarray] = None, rank: int = 0,
                       world: int = 1, exchange: Optional[Callable] = None, band_rows: Optional[int] = None):
        """This rank's share of the map in halo mode: accumulate band by band, start the exchange of the boundary zones
        with the neighbours (non-blocking send / recv: distributed.exchange_halo_start, RCCL on the GPUs), finalise the
        interior rows beside it, then the zones.  ``exchange``: a blocking replacement with exchange_halo's signature."""
        if img is not None:
            self.setImages(img, dem)
        st = self.haloAccumulate(rank, world, band_rows=band_rows)
        if world == 1:
            return self.haloFinish(st)
        torch.cuda.current_stream(self.device).synchronize()      # the slabs are complete before they are sent
        wp = st["wp"]
        shapes = ((3, st["down_rows"], wp), (3, st["up_rows"], wp))
        if exchange is not None:
            return self.haloFinish(st, *exchange(st["send_down"], st["send_up"], shapes[0], shapes[1], rank, world))
        from .distributed import exchange_halo_start
        wait = exchange_halo_start(st["send_down"], st["send_up"], shapes[0], shapes[1], rank, world)
        return self.haloFinish(st, exchange=wait)

    def cropHalo(self, slabs: Sequence[Tuple[Tuple[torch.Tensor, torch.Tensor, torch.Tensor], Tuple[int, int]]]):
        """Assemble per-rank slabs (in rank order) into the final rasters cropped to the input extent: final pixel
        (y, x) is canvas pixel (y + S - s, x + S - s)."""
        halo = self.image_size - self.stride
        h, w = self.dem_shape
        parts = [torch.cat([sl[0][k] for sl in slabs], dim=0) for k in range(3)]
        return tuple(p[halo:halo + h, halo:halo + w].cpu().numpy() for p in parts)
