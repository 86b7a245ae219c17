"""Diagnostic (not a pytest): kernel-level A/B of the f16c conv forms on two BASELINE layer shapes.
    rocprofv3 --kernel-trace --stats ... -- python3 tools/gpu_sw_bench.py      (MSR_F16C_SW=0: the ping-pong kernel, 2: the stream kernel for both)
main = rb4.conv1 of SPADE-512 B=8 (r 64, 1024 -> 512, bias epilogue);  gb = rb5.gb1 (r 128, 128 -> 2 x 512, SPADE epilogue)."""
import sys
import time
import torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import ops
ctx = ops.OpContext()
g = torch.Generator(device="cpu").manual_seed(5)
res = {}
for kind, (B, r, cin, N) in (("main", (8, 64, 1024, 512)), ("gb", (8, 128, 128, 1024))):
    x = torch.randn((B, r + 2, r + 2, cin), generator=g).cuda()
    w = (torch.randn((9, N, cin), generator=g) * 0.01).cuda()
    bias = torch.zeros(N, device="cuda")
    ximg, _ = ops.f16c_activation_image(x)
    wimg, wexp, _ = ops.f16c_weight_image(w)
    kw = {}
    if kind == "gb":
        C = N // 2
        aux = torch.randn((B, r // 2, r // 2, C), generator=g).cuda()
        kw = dict(epilogue=ops.EPI_SPADE, aux=aux, aux_shift=1, mean=torch.zeros(C, device="cuda"),
                  std=torch.ones(C, device="cuda"), out_padded=True, out_mode=4)
    for _ in range(15):
        ops.conv3x3_f16c(ctx, ximg, wimg, wexp, bias, r, **kw)
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        ops.conv3x3_f16c(ctx, ximg, wimg, wexp, bias, r, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    flop = 2.0 * B * r * r * cin * 9 * N
    res[kind] = (dt * 1e3, flop / dt / 1e12)
    print(f"{kind}: {dt * 1e3:.3f} ms per call incl. the output memset, {flop / dt / 1e12:.0f} TF/s", flush=True)
