"""Micro-benchmark of conv_igemm_f32 on the generator's layer shapes (not a pytest).
usage: python tools/gpu_conv_bench.py [S B [name-filter]]  -> TFLOP/s per layer shape, both tiles"""
import sys
import torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import ops

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
only = sys.argv[3] if len(sys.argv) > 3 else ""
ctx = ops.OpContext()
sw = S // 64
shapes = [("rb1.conv", sw, 1024, 1024, 0), ("rb1.gb", sw, 128, 2048, 2), ("rb2.conv", 2 * sw, 1024, 1024, 0),
          ("rb2.gb", 2 * sw, 128, 2048, 2), ("rb3.conv", 4 * sw, 1024, 1024, 0), ("rb4.conv1", 8 * sw, 1024, 512, 0),
          ("rb5.gb1", 16 * sw, 128, 1024, 2), ("rb6.gb1", 32 * sw, 128, 512, 2), ("rb6.conv1", 32 * sw, 256, 128, 0),
          ("rb6.conv2", 32 * sw, 128, 128, 1)]
for name, r, cin, N, epi in shapes:
    if only and only not in name:
        continue
    x = torch.randn((B, r + 2, r + 2, cin), device="cuda")
    w = torch.randn((9, N, cin), device="cuda") * 0.01
    bias = torch.zeros(N, device="cuda")
    C = N // 2 if epi == 2 else N
    aux = torch.randn((B, r, r, C), device="cuda") if epi else None
    mean = torch.zeros(C, device="cuda") if epi == 2 else None
    std = torch.ones(C, device="cuda") if epi == 2 else None
    out = torch.zeros((B, r + 2, r + 2, C), device="cuda") if epi == 2 else torch.empty((B, r, r, C), device="cuda")
    line = f"{name:10s} r={r:4d} Cin={cin:5d} N={N:5d}"
    for tile, prec in ((0, "fp32"), (0, "bf16x3"), (0x40, "bf16x3"), (3, "bf16x3"), (4, "bf16x3"), (5, "bf16x3")):
        try:
            kw = dict(epilogue=epi, aux=aux, mean=mean, std=std, out_padded=(epi == 2), tile=tile, out=out, precision=prec,
                      out_split=(epi == 2 and prec == "bf16x3"))   # as the generator runs it
            for _ in range(2):
                ops.conv3x3(ctx, x, w, bias, r, **kw)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 40
            a.record()
            for _ in range(n):
                ops.conv3x3(ctx, x, w, bias, r, **kw)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / n
            line += f" | {tile:02x}{prec[:2]}: {ms:6.3f} {2.0 * B * r * r * cin * N * 9 / ms / 1e9:6.1f}"
        except ValueError as e:
            line += f" | {tile:02x}{prec[:2]}: n/a"
    print(line, flush=True)
