#!/usr/bin/env python3
"""raster_bench.py — BASELINE config 4: process_full_tiles over a synthetic DEM raster, tile-row sharded.

    python raster_bench.py --rows 4096 --cols 4096 --image-size 256 --stride 32 --batch-size 16
    python raster_bench.py --gpus N --rows 15000 --cols 70000 ...     (starts its own N ranks; also runs as a worker
                                                                       under an outer torch.distributed.run)
    python raster_bench.py --rows 15000 --cols 70000 --image-size 512 --stride 64 --batch-size 8 \
                           --simulate-rank 3 --simulate-world 8 --max-tiles 8     (one GPU takes rank 3's shard of 8)

Every rank pads the raster, takes a contiguous block of 1024-px tile rows (moonsuperresolution_amd.distributed),
runs getPatch/normalize -> generator -> rebuildTile entirely on its GPU, and (with --gather) all-gathers the finished
rows over RCCL.  Reports end-to-end patches/s and 512x512-tile-equivalents/s of generator work, per stage times
of rank 0, and the share of time outside the generator (tiler + stitcher + host).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # the tile loop alternates two generator handles (4 streams in all)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the pool's documented environment (dmabuf IPC only): see bench.py

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
    ap.add_argument("--simulate-rank", type=int, default=-1, help="single process: take this rank's shard ...")
    ap.add_argument("--simulate-world", type=int, default=0, help="... of this many ranks (no process group)")
    ap.add_argument("--no-reference", dest="reference", action="store_false",
                    help="skip the generator-only pass (same calls, same streams, no tiler / stitcher) after the map")
    ap.add_argument("--halo", action="store_true",
                    help="patch-row-sharded mode (halo.py): every patch position generated once, neighbour exchange of "
                         "the boundary-zone accumulators over send / recv; NOT reference-identical (see halo.py)")
    ap.add_argument("--max-rows", type=int, default=0, help="halo mode: stop after this many patch rows of the shard (0 = all)")
    ap.add_argument("--band-rows", type=int, default=0, help="halo mode: patch rows per band (0 = what fits --band-gib of predictions)")
    ap.add_argument("--band-gib", type=float, default=2.0, help="halo mode: predictions kept at a time, GiB")
    ap.add_argument("--passes", type=int, default=1, help="run the shard this many times; the last pass is reported")
    ap.add_argument("--dump", default="", help="rank 0 writes the finished products (mean, std, good) to this .npz: the gathered "
                                               "rows with --gather, else its own rows (tests compare them across process counts)")
    ap.add_argument("--precision", default="f16c")
    ap.add_argument("--pipeline", type=int, default=2)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the N ranks as fresh child processes before this one touches HIP
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        raise SystemExit(subprocess.call(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
             "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist
    from moonsuperresolution_amd import DEMSuperResolution, DSRConfig, Generator
    from moonsuperresolution_amd.distributed import all_gather_rows, shard_tile_rows, tile_rows

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # as bench.py: MSR_BENCH_BACKEND=gloo (+ MSR_BENCH_DEVICE) rehearses the N-rank path on a one-GPU box
    backend = os.environ.get("MSR_BENCH_BACKEND", "nccl")
    if "MSR_BENCH_DEVICE" in os.environ:
        local = int(os.environ["MSR_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    S, s, B, T = args.image_size, args.stride, args.batch_size, args.tile_size
    t_setup = time.perf_counter()
    img, dem = synthetic_raster(args.rows, args.cols, seed=0)
    gen = Generator(S, B, variant="gaugan", weights=1234, eps=7, device=local, precision=args.precision)
    dsr = DEMSuperResolution(DSRConfig(image_size=S, stride=s, batch_size=B, tile_size=T), model=gen, device=local,
                             pipeline=args.pipeline)
    dsr.setImages(img, dem)
    del img, dem
    dsr.padInputs()
    t_setup = time.perf_counter() - t_setup
    tiles = dsr.generateTileList()
    shard_rank, shard_world = rank, world
    if args.simulate_world > 0:
        shard_rank, shard_world = args.simulate_rank, args.simulate_world
    mine = shard_tile_rows(tiles, shard_rank, shard_world)
    if args.halo:
        from moonsuperresolution_amd import HaloShardedSuperResolution
        from moonsuperresolution_amd.distributed import all_gather_var_rows, halo_zone_rows
        hs = HaloShardedSuperResolution(DSRConfig(image_size=S, stride=s, batch_size=B, tile_size=T), model=gen,
                                        device=local, pipeline=args.pipeline)
        hs.dem_shape, hs.img_shape = dsr.dem_shape, dsr.img_shape
        hs.dem_padded, hs.img_padded = dsr.dem_padded, dsr.img_padded
        hs.dem_padded_shape, hs.img_padded_shape = dsr.dem_padded_shape, dsr.img_padded_shape
        gen.forward_device(torch.zeros((B, S, S, 2), device="cuda").uniform_(-0.5, 0.5))
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.cuda.reset_peak_memory_stats()
        st = hs.haloAccumulate(shard_rank, shard_world, band_rows=args.band_rows or None,
                               band_bytes=int(args.band_gib * (1 << 30)), max_rows=args.max_rows)
        torch.cuda.synchronize()
        t_acc = time.perf_counter() - t0
        if world > 1:
            # the zone slabs travel while the interior rows are finalised; the wait sits inside haloFinish
            from moonsuperresolution_amd.distributed import exchange_halo_start
            wait = exchange_halo_start(st["send_down"], st["send_up"], (3, st["down_rows"], st["wp"]),
                                       (3, st["up_rows"], st["wp"]), rank, world)
            (m, sd, g), (own_lo, own_hi) = hs.haloFinish(st, exchange=wait)
        else:
            (m, sd, g), (own_lo, own_hi) = hs.haloFinish(st)
        torch.cuda.synchronize()
        t_ex = time.perf_counter() - t0 - t_acc
        free_b, total_b = torch.cuda.mem_get_info()
        if world > 1 and args.gather:
            ys, _ = hs.patchGrid()
            zones = halo_zone_rows(ys, S, world)
            counts = [(hs.dem_padded_shape[0] if z["own_hi"] is None else z["own_hi"]) - z["own_lo"] for z in zones]
            m, sd, g = (all_gather_var_rows(t, counts) for t in (m, sd, g))
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if args.dump and rank == 0:
            np.savez(args.dump, mean=m.cpu().numpy(), std=sd.cpu().numpy(), good=g.cpu().numpy(), own=np.array([own_lo, own_hi]))
        nv, nc = hs.last_counts_halo
        tot = torch.tensor([nv, nc], dtype=torch.float64, device="cuda")
        mx = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({
                "metric": "raster end-to-end, halo mode (every patch position generated once; not reference-identical)",
                "n_gpus": world, "raster": [args.rows, args.cols], "image_size": S, "stride": s, "batch_size": B,
                "shard": [shard_rank, shard_world], "patches": int(tot[0]), "generator_calls": int(tot[1]),
                "seconds": float(mx[0]), "seconds_accumulate_rank0": t_acc, "seconds_exchange_and_finalize_rank0": t_ex,
                "band_rows": hs.last_band_rows, "max_rows": args.max_rows,
                "peak_torch_allocated_gb_rank0": torch.cuda.max_memory_allocated() / 1e9,
                "device_memory_in_use_gb_rank0": (total_b - free_b) / 1e9,
                "patches_per_s": float(tot[0]) / float(mx[0]),
                "tiles512_per_s": float(tot[1]) * B * (S / 512.0) ** 2 / float(mx[0]),
                "own_rows_rank0": [own_lo, own_hi], "zone_bytes_sent_rank0":
                    sum(int(t.numel()) * 4 for t in (st["send_down"], st["send_up"]) if t is not None),
                "canvas": list(hs.dem_padded_shape), "precision": args.precision, "pipeline": args.pipeline}))
        if world > 1:
            dist.destroy_process_group()
        return
    todo = mine[:args.max_tiles] if args.max_tiles else mine
    dev = torch.device("cuda", local)
    rows_all = tile_rows(tiles)
    my_rows = tile_rows(mine) if mine else []
    width = len({xx for xx, _ in tiles}) * T
    # this rank's finished rows stay on the GPU: [rows * T, width] per product
    prod = [torch.zeros((len(my_rows) * T, width), dtype=torch.float32, device=dev),
            torch.zeros((len(my_rows) * T, width), dtype=torch.float32, device=dev),
            torch.zeros((len(my_rows) * T, width), dtype=torch.uint8, device=dev)]
    patches = calls = 0
    if todo:
        dsr.processTile(*todo[0])                                     # and one whole tile (clone handles, streams)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    pass_seconds = []
    for _ in range(max(1, args.passes)):          # the LAST pass is the one reported (the first also pays the clock ramp)
        patches = calls = 0
        t0 = time.perf_counter()
        for (xx, yy), (m, sd, g) in dsr.iterTiles(todo):
            nv, nc = dsr.last_counts
            calls += nc
            patches += nv
            r0 = my_rows.index(yy) * T
            prod[0][r0:r0 + T, xx:xx + T] = m
            prod[1][r0:r0 + T, xx:xx + T] = sd
            prod[2][r0:r0 + T, xx:xx + T] = g
        torch.cuda.synchronize()
        t_tiles = time.perf_counter() - t0
        pass_seconds.append(t_tiles)
    t_gather = 0.0
    if world > 1 and args.gather:
        tg = time.perf_counter()
        prod = [all_gather_rows(t, len(rows_all), T, world) for t in prod]
        torch.cuda.synchronize()
        t_gather = time.perf_counter() - tg
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if args.dump and rank == 0:
        np.savez(args.dump, mean=prod[0].cpu().numpy(), std=prod[1].cpu().numpy(), good=prod[2].cpu().numpy())
    good_fraction = float(prod[2][:, :args.cols].float().mean()) if prod[2].numel() else 0.0
    # generator-only references on rank 0's device: the same number of calls over the same handles / streams, no tiler
    # and no stitcher — (a) on the patches the tile loop left in its batch buffers (real data: clocks under load depend
    # on the operand values), (b) on uniform noise
    gen_only = gen_only_noise = None
    if args.reference and calls and dsr._gens:
        def gen_pass(inputs):
            ob = [torch.empty((B, S, S, 1), device="cuda") for _ in dsr._gens]
            torch.cuda.synchronize()
            tr = time.perf_counter()
            cur = torch.cuda.current_stream()
            for ps in dsr._pstreams:
                ps.wait_stream(cur)
            for c in range(calls):
                k = c % len(dsr._gens)
                with torch.cuda.stream(dsr._pstreams[k]):
                    dsr._gens[k].forward_device(inputs[k], out=ob[k])
            torch.cuda.synchronize()
            return calls * B * (S / 512.0) ** 2 / (time.perf_counter() - tr)
        # a FULL batch of real patches (the loop leaves the tile's last, zero-padded batch in the buffers; zeros run at a
        # higher clock): the first B valid patches of the first tile of the shard
        st0 = dsr._prepare_tile(*todo[0])
        st0["event"].synchronize()
        real = torch.empty((B, S, S, 2), device="cuda")
        rows_p, cols_p = dsr.dem_padded_shape
        dsr._lib.msr_extract_patches(dsr._h, dsr.img_padded.data_ptr(), dsr.dem_padded.data_ptr(), rows_p, cols_p,
                                     st0["sx"].data_ptr(), st0["sy"].data_ptr(), st0["mm_sel"].data_ptr(), B,
                                     real.data_ptr(), torch.cuda.current_stream().cuda_stream)
        gen_only = gen_pass([real for _ in dsr._gens])
        gen_only_noise = gen_pass([torch.zeros((B, S, S, 2), device="cuda").uniform_(-0.5, 0.5) for _ in dsr._gens])
    tot = torch.tensor([patches, calls, elapsed], dtype=torch.float64, device="cuda")
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
            "seconds_tiles_rank0": t_tiles, "seconds_gather_rank0": t_gather, "seconds_per_pass_rank0": pass_seconds,
            "patches_per_s": patches / elapsed,
            "tiles512_per_s": calls * B * (S / 512.0) ** 2 / elapsed,
            "generator_only_tiles512_per_s_rank0": gen_only, "generator_only_on_noise_tiles512_per_s_rank0": gen_only_noise,
            "end_to_end_over_generator_only": (calls * B * (S / 512.0) ** 2 / t_tiles / gen_only) if gen_only else None,
            "good_fraction_of_rank0_rows": good_fraction, "gathered": bool(args.gather and world > 1),
            "canvas": list(dsr.dem_padded_shape), "shard": [shard_rank, shard_world], "tiles_this_rank": len(mine),
            "tiles_processed_this_rank": len(todo),
            "setup_seconds_rank0": t_setup, "precision": args.precision, "pipeline": args.pipeline,
            "device_mem_gib_torch_peak": torch.cuda.max_memory_allocated() / 2 ** 30,
            "device_mem_gib_generator_handles": args.pipeline * gen.device_bytes() / 2 ** 30,
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
