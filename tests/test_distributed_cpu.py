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


def _worker(rank, world, port, shape, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tiles = [(xx, yy) for yy in range(0, shape[0], T) for xx in range(0, shape[1], T)]
    out = process_map_sharded(shape, T, tiles, lambda x, y: _fake_tile(x, y, T), rank, world, gather=True)
    q.put((rank, [o.copy() for o in out]))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,shape", [
    (2, (200, 150)),      # 4 tile rows x 3 tile columns; rows split 2 / 2
    (3, (250, 100)),      # 4 tile rows on 3 ranks: 2 / 1 / 1 (two short ranks: the gathered blocks are compacted)
    (3, (300, 70)),       # 5 tile rows on 3 ranks: 2 / 2 / 1
])
def test_sharded_map_equals_single_process(world, shape):
    T = 64
    tiles = [(xx, yy) for yy in range(0, shape[0], T) for xx in range(0, shape[1], T)]
    single = process_map_sharded(shape, T, tiles, lambda x, y: _fake_tile(x, y, T), 0, 1)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        for a, b in zip(results[r], single):
            assert a.shape == shape and np.array_equal(a, b)


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
