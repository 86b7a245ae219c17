import sys, time, statistics, torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import Generator, synthetic_patches
for B in (1, 16):
    gen = Generator(256, B, variant="pix2pix", weights=1234)
    x = torch.from_numpy(synthetic_patches(B, 256, 1)).cuda()
    out = torch.empty((B, 256, 256, 1), device="cuda")
    for _ in range(3): gen.forward_device(x, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); gen.forward_device(x, out=out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"pix2pix B={B}: p50 {statistics.median(ts):.3f} ms per call, {gen.forward_flops()/1e9:.1f} GFLOP -> {gen.forward_flops()/statistics.median(ts)/1e9:.2f} TFLOP/s", flush=True)
    gen.profile(True); gen.forward_device(x, out=out); print({k: round(v['device_ms'],3) for k,v in gen.profile_read().items()}); gen.profile(False)
    gen.close(); del gen
