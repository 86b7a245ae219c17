"""Probe (not a pytest): throughput of K generator calls issued on ONE stream vs alternated over TWO handles / streams."""
import sys, time, torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import Generator, make_weights, make_latent_noise, synthetic_patches
S, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 16)
w = make_weights("gaugan", S, seed=1234)
eps = make_latent_noise(B, 256, 7)
gens = [Generator(S, B, weights=w, eps=eps) for _ in range(2)]
xs = [torch.from_numpy(synthetic_patches(B, S, seed=i)).cuda() for i in range(2)]
outs = [torch.empty((B, S, S, 1), device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
def run(n, two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i % 2 if two else 0
        with torch.cuda.stream(streams[k]):
            gens[k].forward_device(xs[k], out=outs[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(2):
    run(10, False); run(10, True)
for rep in range(3):
    a = run(60, False); b = run(60, True)
    tiles = B * (S / 512) ** 2
    print(f"one stream {a:.3f} ms/call ({tiles / a * 1e3:.1f} tiles/s) | two streams {b:.3f} ms/call ({tiles / b * 1e3:.1f} tiles/s)  x{a / b:.3f}", flush=True)
