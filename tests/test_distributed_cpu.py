"""CPU: the N>1 path (tile-row sharding + all_gather of finished rows) with world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from moonsuperresolution_amd.distributed import process_map_sharded


def _fake_tile(xx, yy, T=64):
    rng = np.random.default_rng([xx, yy])
    return (rng.standard_normal((T, T)).astype(np.float32), rng.uniform(0, 1, (T, T)).astype(np.float32),
            (rng.uniform(0, 1, (T, T)) > 0.2).astype(np.uint8))


def _smooth_model(x, training=False):
    """A non-trivial stand-in generator (mixes the two channels and neighbouring pixels): the stitched result depends on
    every patch and on its position, unlike the identity model."""
    x = np.asarray(x, np.float32)
    y = 0.6 * x[..., 1:2] + 0.4 * x[..., 0:1]
    y[:, 1:, 1:] += 0.05 * y[:, :-1, :-1]
    return y


def _oracle_tile_fn(shape, S, s, B, T):
    """process_tile(xx, yy) of the ORACLE's real tile path (pad -> patches -> validity -> normalise -> batches ->
    model -> rebuildTile, oracle/tiler_ref.py::process_tile = process_full_tiles.py:431-479) on a synthetic raster."""
    from oracle import tiler_ref
    from tests.helpers import synthetic_raster
    img, dem = synthetic_raster(shape[0], shape[1], seed=11, hole=(30, 70, 40, 90))
    img_p, dem_p = tiler_ref.pad_inputs(img, dem, S, s, -32768.0)
    return lambda xx, yy: tiler_ref.process_tile(img_p, dem_p, xx, yy, _smooth_model, S, s, B, T, -32768.0)


def _worker(rank, world, port, shape, T, q, real):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tiles = [(xx, yy) for yy in range(0, shape[0], T) for xx in range(0, shape[1], T)]
    fn = _oracle_tile_fn(shape, 32, 8, 5, T) if real else (lambda x, y: _fake_tile(x, y, T))
    out = process_map_sharded(shape, T, tiles, fn, rank, world, gather=True)
    q.put((rank, [o.copy() for o in out]))
    dist.barrier()
    dist.destroy_process_group()


import pytest


def _run_world(world, shape, T, real):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, T, q, real)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("world,shape", [
    (2, (200, 150)),      # 4 tile rows x 3 tile columns; rows split 2 / 2
    (3, (250, 100)),      # 4 tile rows on 3 ranks: 2 / 1 / 1 (two short ranks: the gathered blocks are compacted)
    (3, (300, 70)),       # 5 tile rows on 3 ranks: 2 / 2 / 1
])
def test_sharded_map_equals_single_process(world, shape):
    T = 64
    tiles = [(xx, yy) for yy in range(0, shape[0], T) for xx in range(0, shape[1], T)]
    single = process_map_sharded(shape, T, tiles, lambda x, y: _fake_tile(x, y, T), 0, 1)
    results = _run_world(world, shape, T, real=False)
    for r in range(world):
        for a, b in zip(results[r], single):
            assert a.shape == shape and np.array_equal(a, b)


@pytest.mark.parametrize("world,shape", [(2, (200, 150)), (3, (250, 100))])
def test_sharded_real_tile_path_equals_the_oracle_map(world, shape):
    """The REAL tile function in separate processes: every rank runs the oracle's process_tile (patch cutting, validity,
    normalisation, zero-padded last batch, stitcher) on its tile rows; the gathered map equals oracle process_map of the
    whole raster bit for bit (process_full_tiles.py:313-325,431-479 sharded by tile row)."""
    from oracle import tiler_ref
    from tests.helpers import synthetic_raster
    T, S, s, B = 64, 32, 8, 5
    img, dem = synthetic_raster(shape[0], shape[1], seed=11, hole=(30, 70, 40, 90))
    want = tiler_ref.process_map(img, dem, _smooth_model, S, s, B, T, -32768.0)
    results = _run_world(world, shape, T, real=True)
    for r in range(world):
        for a, b in zip(results[r], want):
            assert a.shape == shape and np.array_equal(a, b, equal_nan=True)
    assert want[2].any() and not want[2].all()


def test_uneven_rows_three_ranks_layout():
    # 5 tile rows on 3 ranks -> 2,2,1 ; without gather each rank fills only its rows
    shape, T = (5 * 32, 64), 32
    tiles = [(xx, yy) for yy in range(0, shape[0], T) for xx in range(0, shape[1], T)]
    full = process_map_sharded(shape, T, tiles, lambda x, y: _fake_tile(x, y, T), 0, 1)
    acc = [np.zeros_like(f) for f in full]
    for r in range(3):
        part = process_map_sharded(shape, T, tiles, lambda x, y: _fake_tile(x, y, T), r, 3, gather=False)
        for a, p in zip(acc, part):
            a += p
    for a, f in zip(acc, full):
        assert np.array_equal(a, f)


# ---- halo mode: the neighbour exchange and the variable-row gather over gloo, on the oracle's accumulators ----
def _halo_worker(rank, world, port, q, overlapped=False):
    from oracle import tiler_ref
    from moonsuperresolution_amd.distributed import all_gather_var_rows, exchange_halo, exchange_halo_start, halo_zone_rows
    from tests.helpers import synthetic_raster
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, s, B, T, noval = 64, 16, 4, 128, -32768.0
    img, dem = synthetic_raster(260, 150, 31, hole=(100, 130, 40, 90))
    model = lambda x, training=False: np.asarray(x, np.float32)   # noqa: E731
    img_p, dem_p = tiler_ref.pad_inputs(img, dem, S, s, noval)
    ys, xs = tiler_ref.halo_grid(dem.shape, S, s, T)
    zones = halo_zone_rows(ys, S, world)
    z = zones[rank]
    hp, wp = dem_p.shape
    own_lo, own_hi = z["own_lo"], hp if z["own_hi"] is None else z["own_hi"]
    acc = np.stack(tiler_ref.halo_partials(img_p, dem_p, ys[z["g0"]:z["g1"]], xs, model, S, s, B, noval))   # [3, hp, wp]
    lo, hi = z["touch_lo"], z["touch_hi"]
    send_down = torch.from_numpy(acc[:, lo:own_lo].copy()) if rank > 0 else None
    send_up = torch.from_numpy(acc[:, own_hi:hi].copy()) if rank < world - 1 else None
    down_rows = zones[rank - 1]["touch_hi"] - own_lo if rank > 0 else 0
    up_rows = own_hi - zones[rank + 1]["touch_lo"] if rank < world - 1 else 0
    if overlapped:     # the non-blocking form halo.py uses: start, do the rank's own work, wait
        wait = exchange_halo_start(send_down, send_up, (3, down_rows, wp), (3, up_rows, wp), rank, world)
        mine = [a[own_lo:own_hi].copy() for a in acc]
        from_down, from_up = wait()
    else:
        from_down, from_up = exchange_halo(send_down, send_up, (3, down_rows, wp), (3, up_rows, wp), rank, world)
        mine = [a[own_lo:own_hi].copy() for a in acc]
    if from_down is not None:
        m = tiler_ref.chan_merge(tuple(from_down.numpy()), tuple(a[:down_rows] for a in mine))
        for a, b in zip(mine, m):
            a[:down_rows] = b
    if from_up is not None:
        n = own_hi - own_lo
        m = tiler_ref.chan_merge(tuple(a[n - up_rows:] for a in mine), tuple(from_up.numpy()))
        for a, b in zip(mine, m):
            a[n - up_rows:] = b
    out = tiler_ref.halo_finalize(mine, noval)
    counts = [(hp if zz["own_hi"] is None else zz["own_hi"]) - zz["own_lo"] for zz in zones]
    full = [all_gather_var_rows(torch.from_numpy(np.ascontiguousarray(o)), counts).numpy() for o in out]
    halo = S - s
    h, w = dem.shape
    q.put((rank, [f[halo:halo + h, halo:halo + w].copy() for f in full]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,overlapped", [(2, False), (3, False), (3, True)])
def test_halo_exchange_over_gloo_equals_oracle(world, overlapped):
    """exchange_halo (send / recv of the boundary zones; overlapped = exchange_halo_start ... wait) + all_gather_var_rows on
    `world` gloo ranks reproduce the oracle's halo mode run in one process with the same number of simulated ranks, bit
    for bit."""
    from oracle import tiler_ref
    from tests.helpers import synthetic_raster
    img, dem = synthetic_raster(260, 150, 31, hole=(100, 130, 40, 90))
    model = lambda x, training=False: np.asarray(x, np.float32)   # noqa: E731
    ref = tiler_ref.process_map_halo(img, dem, model, 64, 16, 4, 128, -32768.0, world=world)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q, overlapped)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        for a, b in zip(results[r], ref):
            assert np.array_equal(a, b, equal_nan=True)
