"""Diagnostic (not a pytest): the stream kernel on rb4.conv1's shape with fp8 (f16c) and fp6 (f16c6) cross pieces."""
import sys
import time
import torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import ops
ctx = ops.OpContext()
g = torch.Generator(device="cpu").manual_seed(5)
B, r, cin, N = 8, 64, 1024, 512
x = torch.randn((B, r + 2, r + 2, cin), generator=g).cuda()
w = (torch.randn((9, N, cin), generator=g) * 0.01).cuda()
bias = torch.zeros(N, device="cuda")
ximg, _ = ops.f16c_activation_image(x)
wimg, wexp, _ = ops.f16c_weight_image(w)
x6, _ = ops.f16c6_activation_image(x)
w6, _ = ops.f16c6_weight_image(w)
for name, args in (("fp8 pieces", (ximg, wimg, wexp)), ("fp6 pieces", (x6, w6, None)), ("fp8 pieces", (ximg, wimg, wexp)), ("fp6 pieces", (x6, w6, None))):
    for _ in range(15):
        ops.conv3x3_f16c(ctx, args[0], args[1], args[2], bias, r)
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        ops.conv3x3_f16c(ctx, args[0], args[1], args[2], bias, r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt * 1e3:.3f} ms per call incl. the output memset, {2.0 * B * r * r * cin * 9 * N / dt / 1e12:.0f} TF/s", flush=True)
