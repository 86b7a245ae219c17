import sys, time, statistics, torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import Generator, synthetic_patches
for S, B in ((512, 1), (256, 1), (512, 8)):
    gen = Generator(S, B, weights=1234, eps=7)
    x = torch.from_numpy(synthetic_patches(B, S, 1)).cuda()
    out = torch.empty((B, S, S, 1), device="cuda")
    for _ in range(3): gen.forward_device(x, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); gen.forward_device(x, out=out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"S={S} B={B}: p50 single-call latency {statistics.median(ts):.3f} ms  (min {min(ts):.3f})", flush=True)
    gen.close(); del gen
