"""Probe (not a pytest): 64x64 vs 128x128 tiles (weights in VGPRs) with split-K on the low-resolution layer shapes."""
import sys, torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import ops
ctx = ops.OpContext()
shapes = [(16, 8, 1024, 1024, 0), (16, 4, 1024, 1024, 0), (16, 8, 128, 2048, 2), (16, 4, 128, 2048, 2),
          (8, 16, 1024, 1024, 0), (8, 8, 1024, 1024, 0), (8, 8, 128, 2048, 2), (16, 16, 256, 512, 0), (16, 8, 512, 512, 0)]
for B, r, cin, N, epi in shapes:
    x = torch.randn((B, r + 2, r + 2, cin), device="cuda")
    w = torch.randn((9, N, cin), device="cuda") * 0.01
    wf = ops.weights_bf16x3(w)
    xs = ops.split_bf16(ctx, x)
    bias = torch.zeros(N, device="cuda")
    C = N // 2 if epi == 2 else N
    aux = torch.randn((B, r, r, C), device="cuda") if epi else None
    mean = torch.zeros(C, device="cuda") if epi == 2 else None
    std = torch.ones(C, device="cuda") if epi == 2 else None
    out = torch.zeros((B, r + 2, r + 2, C), device="cuda") if epi == 2 else torch.empty((B, r, r, C), device="cuda")
    line = f"B={B:2d} r={r:2d} Cin={cin:4d} N={N:4d} epi={epi}"
    for tile in (0x41, 0x40):
        for ks in (1, 2, 4, 8, 16):
            if 9 * (cin // 32) // (ks * 2) < 4:
                continue
            t = tile | (ks << 8)
            kw = dict(epilogue=epi, aux=aux, mean=mean, std=std, out_padded=(epi == 2), tile=t, out=out,
                      precision="bf16x3", out_split=(epi == 2))
            try:
                for _ in range(3): ops.conv3x3(ctx, xs, wf, bias, r, **kw)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(30): ops.conv3x3(ctx, xs, wf, bias, r, **kw)
                b.record(); torch.cuda.synchronize()
                line += f" | {'64' if tile == 0x41 else '128'}k{ks}:{a.elapsed_time(b) / 30 * 1e3:5.0f}"
            except ValueError:
                line += f" | {'64' if tile == 0x41 else '128'}k{ks}: n/a"
    print(line, flush=True)
