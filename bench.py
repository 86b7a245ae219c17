#!/usr/bin/env python3
"""bench.py — throughput of the tiled DEM super-resolution hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload spade256|spade512]

Conv arithmetic (--precision): "bf16x3" (default) = 3-term split-bf16 products on v_mfma_f32_32x32x16_bf16 with fp32
accumulation, inputs / outputs / weights / all other ops fp32 — 2e-5 relative L-inf vs the float64 oracle, inside the
1e-3 bar of BASELINE.json (tests/test_gpu_generator.py); "fp32" = exact fp32 MFMA (4e-6).

A "step" is one generator(call) — `self.model(np.array(batch), training=False)` of process_full_tiles.py:338 —
over one batch of synthetic (ortho, low-res DEM) patches that is already resident in HBM:
    spade256 (default, BASELINE.json configs[1]): GauGAN(256, 16, 256), batch [16,256,256,2] = 4 tiles of 512x512
    spade512 (configs[2], the MFMA roofline run):  GauGAN(512,  8, 256), batch [ 8,512,512,2] = 8 tiles of 512x512
Metric: 512x512 DEM tiles/s over the whole job (a 512x512 tile = four 256x256 patches).
N > 1: one process per GPU (torch.distributed / RCCL only for the timing barrier + max); patches are
independent units, so ranks share nothing on the data path ("weak" scaling: per-GPU work is fixed).
--streams 2 issues consecutive steps (independent batches) alternately on two generator handles, each on its own HIP
stream: the latency-bound head of one call (encoder, dense, the r <= 8 layers: ~20 % of a call at low occupancy) then
overlaps the matrix-bound tail of the other (+5-7 % throughput; the driver's tile loop, tiler.py, does this by
default).  The default is ONE stream, so that the per-launch durations bench.py reports and a rocprofv3 kernel trace
of the same command agree (overlapping kernels of two streams stretch each other's traced durations).
`p50_ms_per_call` / `p50_latency_ms_per_tile` come from a single-stream pass after the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # 2 call streams + 2 auxiliary streams + torch's own: keep them on separate queues
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    "spade256": dict(S=256, B=16, name="SPADE-256 GauGAN(256,16,256) batch=16 256x256 patches (=4 512x512 tiles) per step"),
    "spade512": dict(S=512, B=8, name="SPADE-512 GauGAN(512,8,256) batch=8 512x512 tiles per step"),
}
# MI355X_MICROARCH.md: dense MFMA peaks.  For bf16x3 every algorithmic product costs three bf16 MFMA products, so
# the algorithmic ceiling is a third of the bf16 peak; `achieved` stays ALGORITHMIC FLOP/s in both cases.
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0}


def host_threads() -> int:
    """Threads for the CPU baseline: the affinity mask, capped by the cgroup CPU quota and by the GPU box's
    per-GPU CPU share (16; override with MSR_CPU_THREADS) — oversubscribing the share only slows the oracle."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("MSR_CPU_THREADS", "16"))))


def pmc_traffic(workload: str):
    """HBM bytes per conv_igemm_f32 launch from the committed rocprofv3 PMC passes (profiles/*_pmc_summary.json,
    produced by profiles/summarize_pmc.py: separate FETCH_SIZE / WRITE_SIZE passes, KiB units, FETCH doubled on
    gfx950 as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside this process."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") == workload:
            best = (d, os.path.basename(f))
    return best


def cpu_baseline(S: int, sample_patches: int, weights, eps_full):
    """Time the oracle (CPU restatement of the reference generator, PyTorch-CPU fp32) on a bounded sample."""
    import numpy as np
    import torch
    from moonsuperresolution_amd import synthetic_patches
    from oracle import generator_ref
    cores = host_threads()
    torch.set_num_threads(cores)
    wt = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
    x = synthetic_patches(sample_patches, S, seed=100)
    generator_ref.spade_call(x[:1], wt, "gaugan", eps_full[:1], dtype=torch.float32)     # warm the thread pool
    t0 = time.perf_counter()
    generator_ref.spade_call(x, wt, "gaugan", eps_full[:sample_patches], dtype=torch.float32)
    dt = time.perf_counter() - t0
    tiles = sample_patches * (S / 512.0) ** 2
    return dict(value=tiles / dt, unit="512x512 tiles/s", cores=cores, kind="port",
                sample=f"one oracle call (PyTorch-CPU fp32 restatement, not TensorFlow) on {sample_patches} "
                       f"patches of {S}x{S} = {tiles:g} tiles, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="spade256")
    ap.add_argument("--precision", choices=["fp32", "bf16x3"], default="bf16x3",
                    help="conv arithmetic: exact fp32 MFMA, or 3-term split-bf16 on the bf16 MFMA (fp32 accumulate)")
    ap.add_argument("--streams", type=int, default=1, help="generator handles / HIP streams the steps alternate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    # Rehearsal knobs (one-GPU boxes only): MSR_BENCH_DEVICE pins every rank to one device and MSR_BENCH_BACKEND=gloo
    # replaces RCCL, so the N>1 plumbing can be exercised without N GPUs.  The driver's runs set neither.
    if "MSR_BENCH_DEVICE" in os.environ:
        local = int(os.environ["MSR_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    backend = os.environ.get("MSR_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from moonsuperresolution_amd import Generator, make_latent_noise, make_weights, synthetic_patches

    wl = WORKLOADS[args.workload]
    S, B = wl["S"], wl["B"]
    weights = make_weights("gaugan", S, seed=1234)
    eps = make_latent_noise(B, 256, seed=7)
    ns = max(1, args.streams)
    gens = [Generator(S, B, variant="gaugan", weights=weights, eps=eps, device=local, precision=args.precision)
            for _ in range(ns)]
    gen = gens[0]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    # a small pool of distinct synthetic batches, resident in HBM before the timed region
    pool = [torch.from_numpy(synthetic_patches(B, S, seed=1000 * rank + i)).cuda() for i in range(2)]
    outs = [torch.empty((B, S, S, 1), dtype=torch.float32, device="cuda") for _ in range(ns)]
    out = outs[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, ns)):
        with torch.cuda.stream(streams[i % ns]):
            gens[i % ns].forward_device(pool[i % len(pool)], out=outs[i % ns])
    barrier()
    for g in gens:
        g.profile(0 if args.no_profile else 2)    # dominant kernel family only, one event pair per run of launches
    ref = torch.cuda.Event(enable_timing=True)
    ref.record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        with torch.cuda.stream(streams[i % ns]):
            gens[i % ns].forward_device(pool[i % len(pool)], out=outs[i % ns])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = {}
    if not args.no_profile:
        for g in gens:
            for k, v in g.profile_read().items():
                acc = stats.setdefault(k, dict(launches=0, device_ms=0.0, flops=0.0, bytes=0.0))
                for f in acc:
                    acc[f] += v[f]
    # With several streams the conv intervals of different handles overlap (a run that starts while the other stream's
    # kernel holds the CUs includes its wait), so the family's time is the UNION of the intervals, not their sum.
    conv_union_ms = 0.0
    if not args.no_profile:
        runs = sorted(r[:2] for g in gens for r in g.profile_runs(ref))
        cur_a, cur_b = None, None
        for a, b in runs:
            if cur_b is None or a > cur_b:
                if cur_b is not None:
                    conv_union_ms += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        if cur_b is not None:
            conv_union_ms += cur_b - cur_a
    for g in gens:
        g.profile(False)
    for o in outs:
        assert torch.isfinite(o).all(), "non-finite generator output"
    # latency of one call with nothing else in flight (separate single-stream pass, after the timed region)
    lat_n = 10
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(lat_n)]
    for i in range(lat_n):
        ev[i][0].record()
        gen.forward_device(pool[i % len(pool)], out=out)
        ev[i][1].record()
    torch.cuda.synchronize()
    step_ms = [a.elapsed_time(b) for a, b in ev]
    all_stats, extra = {}, 3
    if not args.no_profile:
        # every family, every launch bracketed: a separate short pass AFTER the timed region (it costs ~7 %)
        gen.profile(1)
        for i in range(extra):
            gen.forward_device(pool[i % len(pool)], out=out)
        all_stats = gen.profile_read()
        gen.profile(False)

    tiles_per_step = B * (S / 512.0) ** 2
    value = world * args.steps * tiles_per_step / elapsed
    if rank == 0:
        res = {
            "metric": "512x512 DEM tiles/s (whole job)", "value": value, "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16x3 (f32 in/out/accumulate)",
            "data": "synthetic",
            "config": {"workload": wl["name"], "image_size": S, "batch_size": B, "variant": "gaugan",
                       "weights": "random-init (Keras default distributions), seed 1234",
                       "tiles_per_step_per_gpu": tiles_per_step, "parallelism": f"tile-sharded x{world}",
                       "streams_per_gpu": ns},
            "patches_per_s": world * args.steps * B / elapsed,
            "p50_latency_ms_per_tile": statistics.median(step_ms) / tiles_per_step,
            "p50_ms_per_call": statistics.median(step_ms),
            "latency_note": f"p50 of {lat_n} calls on one stream with nothing else in flight (after the timed region)",
            "forward_gflop_per_call": gen.forward_flops() / 1e9,
            "achieved_tflops_whole_call": gen.forward_flops() * args.steps / elapsed / 1e12,
            "device_mem_gib": ns * gen.device_bytes() / 2 ** 30,
        }
        kname = "conv_igemm_f32" if args.precision == "fp32" else "conv_igemm_bf16x3"
        conv = stats.get(kname)
        if conv and conv_union_ms > 0:
            ach = conv["flops"] / (conv_union_ms * 1e-3) / 1e12
            peak = PEAK_TFLOPS[args.precision]
            res["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": ach,
                               "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                               "traffic": None, "traffic_unit": "bytes per launch (HBM, PMC)",
                               "launches": conv["launches"],
                               "avg_launch_ms": conv_union_ms / conv["launches"],
                               "family_busy_ms": conv_union_ms, "sum_of_intervals_ms": conv["device_ms"]}
            res["roofline"]["algorithmic_flops_per_launch"] = conv["flops"] / conv["launches"]
            if args.precision == "bf16x3":
                res["roofline"]["mfma_executed_tflops"] = 3 * ach   # three bf16 MFMA products per algorithmic one
                res["roofline"]["frac_of_executed_mfma_peak"] = 3 * ach / peak
                res["roofline"]["note"] = ("frac = algorithmic FLOP/s over the dense bf16 MFMA peak; bf16x3 issues three bf16 "
                                           "MFMA products per algorithmic product, so the matrix pipe executes 3x `achieved` "
                                           "(frac_of_executed_mfma_peak); SQ_VALU_MFMA_BUSY_CYCLES of the dominant kernel: "
                                           "profiles/r01_spade256_bf16x3_sq_counters.txt (80 % long-K, 70 % gamma/beta layers "
                                           "at the ~1.94 GHz the chip holds under this load)")
            pmc = pmc_traffic(args.workload + ("" if args.precision == "fp32" else "_bf16x3"))
            if pmc and "conv_igemm" in pmc[0]:
                res["roofline"]["traffic"] = pmc[0]["conv_igemm"]["hbm_bytes_per_forward"] / (conv["launches"] / args.steps)
                res["roofline"]["traffic_source"] = "profiles/" + pmc[1]
            res["roofline"]["timing"] = ("HIP events on the calls' streams over the timed region, one pair per run of "
                                         "consecutive conv launches (inter-launch gaps of a run included); family time "
                                         "= union of the intervals over the streams")
            res["kernel_ms_per_call"] = {k: v["device_ms"] / extra for k, v in all_stats.items()}
            res["kernel_ms_per_call_note"] = f"separate pass of {extra} calls after the timed region, every launch bracketed"
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(S, B, weights, eps)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
