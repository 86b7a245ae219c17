#!/usr/bin/env python3
"""bench.py — throughput of the tiled DEM super-resolution hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload spade512|spade256] [--precision bf16x3|fp32]

A "step" is one generator(call) — `self.model(np.array(batch), training=False)` of process_full_tiles.py:338 —
over one batch of synthetic (ortho, low-res DEM) patches that is already resident in HBM:
    spade512 (default, BASELINE.json configs[2], the largest single-GPU config = "the MFMA roofline run"):
              GauGAN(512,  8, 256), batch [ 8,512,512,2] = 8 tiles of 512x512 per step
    spade256 (configs[1]): GauGAN(256, 16, 256), batch [16,256,256,2] = 4 tiles of 512x512 per step
Metric: 512x512 DEM tiles/s over the whole job (a 512x512 tile = four 256x256 patches).

Conv arithmetic (--precision; inputs / outputs / weights / accumulation / all other ops fp32 in every mode):
"f16c" (default) = fp16 main term on v_mfma_f32_16x16x32_f16 + both cross terms on the block-scaled fp8 / fp6 MFMA in the convs
that fill the chip (2 / 1.5 MFMA-equivalents per product: main convs / SPADE layers), 3-term split-bf16 elsewhere — 3.6-4.7e-5 relative L-inf vs the
oracle on these very shapes (tests/test_gpu_baseline_configs.py), inside the 1e-3 bar of BASELINE.json; "bf16x3" = 3-term
split-bf16 everywhere (1.7-2.0e-5); "fp32" = exact fp32 MFMA; "bf16x3_gbf16" opt-in; "fp8" declared non-parity.

At N = 1 the one JSON line also carries, under "also", the driver-timed figures of the other modes and of the other
single-GPU configuration (spade512 in fp32 / bf16x3 / bf16x3_gbf16 / fp8, spade256 in f16c / bf16x3 / fp32: value, ms_per_step,
roofline each, same K and W), the B = 1 single-call latency ("p50_ms_per_call_b1"), and the CPU baseline.

N > 1: `python bench.py --gpus N` launches its own N workers (fresh child processes of torch.distributed.run, before
this process touches HIP); under an outer `python -m torch.distributed.run ... bench.py --gpus N` (WORLD_SIZE set) it
is a worker.  One process per GPU, RCCL process group.  Patches are independent units, so ranks share nothing while
generating ("weak" scaling: per-GPU work is fixed); the path's one exchange — the all-gather of the finished rows of
the map (moonsuperresolution_amd.distributed.all_gather_rows, what process_map_sharded ends with) — is inside the
timed region, sized to the rows the timed steps complete (K * B patches x stride^2 unique pixels each at the
recommended stride S/8, mean + std float32 and good uint8).
--streams 2 issues consecutive steps alternately on two generator handles / HIP streams (+5 %; the tile loop, tiler.py, does
this by default).  The default is ONE stream, so that the per-launch durations reported here and a rocprofv3 kernel trace
of the same command agree; the two-stream rate of the headline workload is recorded under also.<workload>_<mode>_2streams.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # 2 call streams + 2 auxiliary streams + torch's own: keep them on separate queues
# Multi-process GPU work on this pool: the host driver supports only dmabuf IPC, and the pool's environment exports
# HSA_ENABLE_IPC_MODE_LEGACY=0 for that reason on the build container and on every GPU box (without it RCCL and cross-process
# tensor sharing fail with "hipIpcGetMemHandle: invalid argument" — the pool operator's environment note, not a measurement of
# ours).  setdefault only restores that documented default for a caller that scrubbed its environment; an explicit value wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import statistics
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    "spade512": dict(S=512, B=8, name="SPADE-512 GauGAN(512,8,256) batch=8 512x512 tiles per step (BASELINE configs[2])"),
    "spade256": dict(S=256, B=16, name="SPADE-256 GauGAN(256,16,256) batch=16 256x256 patches (=4 512x512 tiles) per step "
                                      "(BASELINE configs[1])"),
}
# MI355X_MICROARCH.md: dense MFMA peaks.  For bf16x3 every algorithmic product costs three bf16 MFMA products, so
# the algorithmic ceiling is a third of the bf16 peak; `achieved` stays ALGORITHMIC FLOP/s in every mode.
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "bf16x3_gbf16": 2500.0, "fp8": 5000.0, "f16c": 2500.0, "f16": 2500.0}
# MFMA products per algorithmic product; the opt-in mode runs 2 in the gamma|beta convs (half of the FLOPs), 3 elsewhere
# f16c since round 3: the gamma|beta convs (half of the FLOPs) run conv_gb_resident with fp6 cross terms (1.5), the main convs 2
# f16: one product in conv_gb_resident / conv_igemm_f16c_sw (~95 % of the FLOPs), f16c / bf16x3 forms in the rest
MFMA_PER_PRODUCT = {"fp32": 1, "bf16x3": 3, "bf16x3_gbf16": 2.5, "fp8": 1, "f16c": 1.75, "f16": 1.1}
# the kernels behind the conv family of each mode (the library's profiler names the family by its base mode)
KERNELS = {"fp32": "conv_igemm<..., PREC_F32> (v_mfma_f32_32x32x2_f32)",
           "bf16x3": "conv_igemm_bf16x3_pp (ping-pong, 3-term split-bf16) + small-tile split-K forms",
           "bf16x3_gbf16": "conv_igemm_bf16x3_pp (gamma|beta convs 2-term fp16, main convs 3-term split-bf16)",
           "f16c": "conv_gb_resident (SPADE layers: embedding + gamma|beta conv + epilogue, f16 + fp6 cross terms) + "
                   "conv_igemm_f16c_sw (main convs, f16 + fp8 cross terms); bf16x3 forms on the layers that do not fill the chip",
           "f16": "conv_gb_resident<NOX> + conv_igemm_f16c_sw<NOX> (one v_mfma_f32_16x16x32_f16 product per element, cross terms "
                  "left out); f16c / bf16x3 forms on the layers that do not fill the chip",
           "fp8": "conv_igemm_bf16x3_pp<PP_FP8> (block-scaled v_mfma_scale_f32_16x16x128_f8f6f4, fp8 x fp8); bf16x3 forms elsewhere"}
DTYPE = {"fp32": "f32", "bf16x3": "bf16x3 (f32 in/out/accumulate)",
         "bf16x3_gbf16": "bf16x3, gamma|beta convs f16x2 (opt-in; f32 in/out/accumulate; 2-5e-4 rel L-inf, inside the 1e-3 bar)",
         "f16c": "f16 main term + fp8 / fp6 cross terms in the chip-filling convs, bf16x3 elsewhere (f32 in/out/accumulate; parity mode)",
         "f16": "one fp16 product per element in the two big kernels (DECLARED TOLERANCE, the usable reading of BASELINE configs[4]; "
                "f32 in/out/accumulate; error stated in tests/test_gpu_baseline_configs.py)",
         "fp8": "fp8 e4m3 weights x bf8 e5m2 activations in the chip-filling convs, bf16x3 elsewhere (DECLARED NON-PARITY: "
                "BASELINE configs[4]; f32 in/out/accumulate; error stated in tests/test_gpu_baseline_configs.py)"}


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
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes (profiles/*_pmc_summary.json, produced by
    profiles/summarize_pmc.py: separate FETCH_SIZE / WRITE_SIZE passes, KiB units, FETCH doubled on gfx950 as
    MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside this process.  Newest round wins."""
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


def self_launch(args) -> None:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as fresh child processes (this process
    has not imported torch or touched HIP) and exit with their code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


class Dist:
    """The process group of a run: RCCL ("nccl") on the GPUs; MSR_BENCH_BACKEND=gloo (+ MSR_BENCH_DEVICE) rehearses
    the N > 1 plumbing on a one-GPU box, moving tensors through the host for its collectives."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if "MSR_BENCH_DEVICE" in os.environ:
            self.local = int(os.environ["MSR_BENCH_DEVICE"])
        torch.cuda.set_device(self.local)
        self.backend = os.environ.get("MSR_BENCH_BACKEND", "nccl")
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.local))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max(self, v: float) -> float:
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_rows(self, products, n_rows, T):
        """The path's exchange step: all-gather of the finished rows of every product."""
        from moonsuperresolution_amd.distributed import all_gather_rows
        out = []
        for t in products:
            if self.backend != "nccl":
                t = t.cpu()
            out.append(all_gather_rows(t, n_rows, T, self.world))
        return out

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def run_workload(args, D: Dist, workload: str, precision: str, with_cpu: bool, with_b1: bool):
    """Warm up, time exactly K steps between barriers, and return the result dict of one (workload, precision)."""
    import torch
    from moonsuperresolution_amd import Generator, make_latent_noise, make_weights, synthetic_patches
    wl = WORKLOADS[workload]
    S, B = wl["S"], wl["B"]
    rank, world, local = D.rank, D.world, D.local
    weights = make_weights("gaugan", S, seed=1234)
    eps = make_latent_noise(B, 256, seed=7)
    ns = max(1, args.streams)
    gens = [Generator(S, B, variant="gaugan", weights=weights, eps=eps, device=local, precision=precision)
            for _ in range(ns)]
    gen = gens[0]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    if ns > 1:      # call streams first, then the handles' plans: four consecutive hardware-queue ids (tiler.py _make_pipeline)
        for st in streams:
            with torch.cuda.stream(st):
                torch.zeros(1, device="cuda")
        for g in gens:
            g.prepare()
    # a small pool of distinct synthetic batches, resident in HBM before the timed region
    pool = [torch.from_numpy(synthetic_patches(B, S, seed=1000 * rank + i)).cuda() for i in range(2)]
    outs = [torch.empty((B, S, S, 1), dtype=torch.float32, device="cuda") for _ in range(ns)]
    out = outs[0]
    # N > 1: the finished rows the K timed steps complete (see the module docstring); T-row blocks, one per rank
    stride = S // 8
    T = 1024
    rows_px = args.steps * B * stride * stride
    width = 4 * T
    n_blocks = max(1, -(-rows_px // (width * T)))
    products = None
    if world > 1:
        products = [torch.zeros((n_blocks * T, width), dtype=torch.float32, device="cuda"),
                    torch.zeros((n_blocks * T, width), dtype=torch.float32, device="cuda"),
                    torch.zeros((n_blocks * T, width), dtype=torch.uint8, device="cuda")]
        D.gather_rows(products, n_blocks * world, T)          # warm the communicator (ring set-up, IPC handles)

    for i in range(max(args.warmup, ns)):
        with torch.cuda.stream(streams[i % ns]):
            gens[i % ns].forward_device(pool[i % len(pool)], out=outs[i % ns])
    D.barrier()
    for g in gens:
        g.profile(0 if args.no_profile else 2)    # dominant kernel family only, one event pair per run of launches
    ref = torch.cuda.Event(enable_timing=True)
    ref.record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        with torch.cuda.stream(streams[i % ns]):
            gens[i % ns].forward_device(pool[i % len(pool)], out=outs[i % ns])
    t_gather = 0.0
    if world > 1:
        torch.cuda.synchronize()
        tg = time.perf_counter()
        gathered = D.gather_rows(products, n_blocks * world, T)
        torch.cuda.synchronize()
        t_gather = time.perf_counter() - tg
        assert gathered[0].shape[0] == n_blocks * world * T
    D.barrier()
    elapsed = D.max(time.perf_counter() - t0)
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
    res = {
        "metric": "512x512 DEM tiles/s (whole job)", "value": world * args.steps * tiles_per_step / elapsed,
        "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE[precision], "data": "synthetic",
        "config": {"workload": wl["name"], "image_size": S, "batch_size": B, "variant": "gaugan",
                   "weights": "random-init (Keras default distributions), seed 1234",
                   "tiles_per_step_per_gpu": tiles_per_step, "parallelism": f"tile-sharded x{world}",
                   "streams_per_gpu": ns},
        "patches_per_s": world * args.steps * B / elapsed,
        "p50_latency_ms_per_tile": statistics.median(step_ms) / tiles_per_step,
        "p50_ms_per_call": statistics.median(step_ms),
        "latency_note": f"p50 of {lat_n} calls on one stream with nothing else in flight (after the timed region)",
        "forward_gflop_per_call": gen.forward_flops() / 1e9,
        "achieved_tflops_whole_call": world * gen.forward_flops() * args.steps / elapsed / 1e12,
        "device_mem_gib": ns * gen.device_bytes() / 2 ** 30,
    }
    if world > 1:
        res["exchange"] = {"collective": f"all_gather_into_tensor x3 ({D.backend}) of the finished rows, inside the timed region",
                           "bytes_per_rank": n_blocks * T * width * 9, "seconds_rank0": t_gather}
    kname = "conv_igemm_f32" if precision == "fp32" else "conv_igemm_bf16x3"      # the library names the family by its base mode
    conv = stats.get(kname)
    if conv and conv_union_ms > 0:
        ach = conv["flops"] / (conv_union_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[precision]
        rl = {"bound": "mfma", "kernel": KERNELS[precision], "profiler_family": kname, "achieved": ach, "peak": peak,
              "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
              "traffic_unit": "bytes per launch (memory-side L2 counters of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE; Infinity-Cache "
                              "hits are counted as traffic by these counters)", "launches": conv["launches"],
              "avg_launch_ms": conv_union_ms / conv["launches"], "family_busy_ms": conv_union_ms,
              "sum_of_intervals_ms": conv["device_ms"],
              "algorithmic_flops_per_launch": conv["flops"] / conv["launches"]}
        if MFMA_PER_PRODUCT[precision] > 1:
            k = MFMA_PER_PRODUCT[precision]
            rl["mfma_executed_tflops"] = k * ach
            rl["frac_of_executed_mfma_peak"] = k * ach / peak
            rl["note"] = (f"frac = algorithmic FLOP/s over the dense bf16/f16 MFMA peak; {precision} issues {k} MFMA-"
                          "equivalents per algorithmic product (average over the conv FLOPs), so the matrix pipe executes "
                          f"that multiple of `achieved` (frac_of_executed_mfma_peak) and frac is capped at {1 / k:.3f}")
        # the modes that keep the split-bf16 tensor geometry move the same bytes: fall back to the bf16x3 passes
        pmc = pmc_traffic(workload + ("" if precision == "fp32" else "_" + precision)) or (
            pmc_traffic(workload + "_bf16x3") if precision in ("f16c", "bf16x3_gbf16") else None)
        if pmc and "conv_igemm" in pmc[0]:
            rl["traffic"] = pmc[0]["conv_igemm"]["hbm_bytes_per_forward"] / (conv["launches"] / args.steps)
            rl["traffic_source"] = "profiles/" + pmc[1]
        rl["timing"] = ("HIP events on the calls' streams over the timed region, one pair per run of consecutive conv "
                        "launches (inter-launch gaps of a run included); family time = union of the intervals over the "
                        "streams")
        res["roofline"] = rl
        res["kernel_ms_per_call"] = {k: v["device_ms"] / extra for k, v in all_stats.items()}
        res["kernel_ms_per_call_note"] = f"separate pass of {extra} calls after the timed region, every launch bracketed"
    for g in gens:
        g.close()
    del gens, gen, pool, outs
    torch.cuda.empty_cache()
    if with_b1:
        # single-call latency at B = 1 (the serial call of process_full_tiles.py:338 with one patch in flight)
        g1 = Generator(S, 1, variant="gaugan", weights=weights, eps=eps[:1], device=local, precision=precision)
        x1 = torch.from_numpy(synthetic_patches(1, S, seed=5)).cuda()
        o1 = torch.empty((1, S, S, 1), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()

        def b1_latencies(n):
            lat = []
            for _ in range(n):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                g1.forward_device(x1, out=o1)
                b.record()
                b.synchronize()
                lat.append(a.elapsed_time(b))
            return lat

        # on a stream of its own: the legacy default stream (torch's current stream otherwise) cannot be captured, and
        # msr_forward then stays eager whatever msr_graph_enable says (round 2's "graph" figure was measured that way)
        with torch.cuda.stream(torch.cuda.Stream()):
            for _ in range(5):
                g1.forward_device(x1, out=o1)
            torch.cuda.current_stream().synchronize()
            res["p50_ms_per_call_b1_eager"] = statistics.median(b1_latencies(30))
            g1.use_graph(True)                               # the launch plan as one HIP graph (same buffers every call)
            for _ in range(4):                               # first sighting eager, second captures, then replays
                g1.forward_device(x1, out=o1)
            torch.cuda.current_stream().synchronize()
            res["p50_ms_per_call_b1_graph"] = statistics.median(b1_latencies(30))
        # the figure a caller gets: the faster of the two launch modes (eager is the library's default; graph replay is opt-in
        # and, with ~90 nodes of 5-60 us, costs more per node than the eager launches it replaces on this ROCm)
        eager_faster = res["p50_ms_per_call_b1_eager"] <= res["p50_ms_per_call_b1_graph"]
        res["p50_ms_per_call_b1"] = min(res["p50_ms_per_call_b1_eager"], res["p50_ms_per_call_b1_graph"])
        res["p50_ms_per_call_b1_mode"] = "eager" if eager_faster else "graph"
        res["p50_ms_per_call_b1_note"] = (f"GauGAN({S},1,256): median of 30 single calls, each synchronised, B = 1; _eager = launched "
                                          "kernel by kernel (the default), _graph = launch plan replayed as a HIP graph "
                                          "(msr_graph_enable); p50_ms_per_call_b1 = the faster mode, named in _mode")
        g1.close()
        del g1
        torch.cuda.empty_cache()
    if with_cpu:
        res["cpu_baseline"] = cpu_baseline(S, B, weights, eps)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="spade512")
    ap.add_argument("--precision", choices=sorted(PEAK_TFLOPS), default="f16c",
                    help="conv arithmetic (fp32 accumulation in every mode): f16c = fp16 main term + fp8 cross terms (default, "
                         "parity), bf16x3 = 3-term split-bf16 (parity), fp32 = exact fp32 MFMA (parity), bf16x3_gbf16 = opt-in "
                         "2-term gamma|beta convs, fp8 = declared non-parity")
    ap.add_argument("--streams", type=int, default=1, help="generator handles / HIP streams the steps alternate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the other single-GPU configurations and the B=1 latency")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    D = Dist(args)
    solo = D.world == 1
    res = run_workload(args, D, args.workload, args.precision, with_cpu=solo and not args.no_cpu_baseline,
                       with_b1=solo and not args.no_also)
    if solo and not args.no_also:
        also = {}
        for wl, prec in (("spade512", "fp32"), ("spade512", "bf16x3"), ("spade512", "bf16x3_gbf16"), ("spade512", "fp8"), ("spade512", "f16"),
                         ("spade256", "f16c"), ("spade256", "bf16x3"), ("spade256", "fp32")):
            if (wl, prec) == (args.workload, args.precision):
                continue
            r = run_workload(args, D, wl, prec, with_cpu=False, with_b1=False)
            also[f"{wl}_{prec}"] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config",
                                                       "p50_ms_per_call", "roofline") if k in r}
        if args.streams == 1:
            # the headline workload the way the tile loop runs it (tiler.py pipeline=2): consecutive steps alternate over two
            # handles / streams, the low-occupancy head of one call overlaps the matrix-bound tail of the other (+5 %)
            import copy
            a2 = copy.copy(args)
            a2.streams = 2
            r = run_workload(a2, D, args.workload, args.precision, with_cpu=False, with_b1=False)
            also[f"{args.workload}_{args.precision}_2streams"] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup",
                                                                                    "dtype", "config", "roofline") if k in r}
        res["also"] = also
    if D.rank == 0:
        print(json.dumps(res))
    D.close()


if __name__ == "__main__":
    main()
