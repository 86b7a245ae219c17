#!/usr/bin/env python3
"""raster_bench.py — BASELINE config 4: process_full_tiles over a synthetic DEM raster, tile-row sharded.

    python raster_bench.py --rows 4096 --cols 4096 --image-size 256 --stride 32 --batch-size 16
    python -m torch.distributed.run --nproc-per-node N raster_bench.py --gpus N --rows 15000 --cols 70000 ...

Every rank pads the raster, takes a contiguous block of 1024-px tile rows (moonsuperresolution_amd.distributed),
runs getPatch/normalize -> generator -> rebuildTile entirely on its GPU, and (with --gather) all-gathers the finished
rows over RCCL.  Reports end-to-end patches/s and 512x512-tile-equivalents/s of generator work, per stage times
of rank 0, and the share of time outside the generator (tiler + stitcher + host).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # the tile loop alternates two generator handles (4 streams in all)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def synthetic_raster(rows, cols, seed=0, nodata_border=0, no_value=-32768.0):
    import numpy as np
    rng = np.random.default_rng(seed)
    gy, gx = rows // 64 + 2, cols // 64 + 2
    coarse = rng.uniform(0, 1, (2, gy, gx)).astype(np.float32)
    ys = np.linspace(0, gy - 1.001, rows, dtype=np.float32)
    xs = np.linspace(0, gx - 1.001, cols, dtype=np.float32)
    y0, x0 = ys.astype(np.int64), xs.astype(np.int64)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    out = []
    for c in range(2):
        t = coarse[c]
        f = (t[y0][:, x0] * (1 - fy) * (1 - fx) + t[y0 + 1][:, x0] * fy * (1 - fx)
             + t[y0][:, x0 + 1] * (1 - fy) * fx + t[y0 + 1][:, x0 + 1] * fy * fx)
        out.append(f)
    img = (0.8 * out[0] + 0.2 * rng.uniform(0, 1, (rows, cols)).astype(np.float32)).astype(np.float32)
    dem = (-3000.0 + 2000.0 * out[1]).astype(np.float32)
    if nodata_border:
        dem[:nodata_border] = no_value
        dem[:, :nodata_border] = no_value
    return img, dem


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--cols", type=int, default=2048)
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--stride", type=int, default=32)
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--tile-size", type=int, default=1024)
    ap.add_argument("--max-tiles", type=int, default=0, help="process at most this many tiles per rank (0 = all)")
    ap.add_argument("--gather", action="store_true", help="all_gather the finished rows on every rank")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from moonsuperresolution_amd import DEMSuperResolution, DSRConfig, Generator
    from moonsuperresolution_amd.distributed import process_map_sharded, shard_tile_rows

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    S, s, B, T = args.image_size, args.stride, args.batch_size, args.tile_size
    img, dem = synthetic_raster(args.rows, args.cols, seed=0)
    gen = Generator(S, B, variant="gaugan", weights=1234, eps=7, device=local)
    dsr = DEMSuperResolution(DSRConfig(image_size=S, stride=s, batch_size=B, tile_size=T), model=gen, device=local)
    dsr.setImages(img, dem)
    dsr.padInputs()
    tiles = dsr.generateTileList()
    mine = shard_tile_rows(tiles, rank, world)
    if args.max_tiles:
        keep = set(mine[:args.max_tiles])
        tiles = [t for t in tiles if t in keep or t not in mine]
    counts = {"patches": 0, "calls": 0}

    def run_tile(xx, yy):
        if args.max_tiles and (xx, yy) not in keep:
            z = torch.zeros((T, T), device="cuda")
            return z, z, torch.zeros((T, T), dtype=torch.uint8, device="cuda")
        out = dsr.processTile(xx, yy)
        counts["calls"] += len(dsr.last_calls)
        counts["patches"] += sum(k != (-1, -1) for c in dsr.last_calls for k in c)
        return out

    # warm-up: one generator call
    gen.forward_device(torch.zeros((B, S, S, 2), device="cuda"))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mean, std, good = process_map_sharded((args.rows, args.cols), T, tiles, run_tile, rank, world, gather=args.gather,
                                          device=torch.device("cuda", local))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tot = torch.tensor([counts["patches"], counts["calls"], elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        elapsed = float(mx[2])
    patches, calls = int(tot[0]), int(tot[1])
    if rank == 0:
        print(json.dumps({
            "metric": "raster end-to-end (tiler + generator + stitcher)", "n_gpus": world,
            "raster": [args.rows, args.cols], "image_size": S, "stride": s, "batch_size": B, "tile_size": T,
            "tiles_total": len(tiles), "patches": patches, "generator_calls": calls, "seconds": elapsed,
            "patches_per_s": patches / elapsed,
            "tiles512_per_s": calls * B * (S / 512.0) ** 2 / elapsed,
            "good_fraction": float(np.mean(good)), "gathered": bool(args.gather),
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
