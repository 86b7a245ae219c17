"""Diagnostic (not a pytest): per-tile cycle counts of the ping-pong kernel on a gamma|beta layer shape (stamped build, see
tools/gpu_pp_stamps.py for the recipe; run with MSR_F16C_SW=0).  36 K-steps per tile: compare with 36 x the per-step figure
of the long-K layers (tools/gpu_pp_stamps_f16c.py)."""
import sys
import torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import ops
ctx = ops.OpContext()
B, r, cin, N = 8, 128, 128, 1024
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn((B, r + 2, r + 2, cin), generator=g).cuda()
w = (torch.randn((9, N, cin), generator=g) * 0.01).cuda()
bias = torch.zeros(N, device="cuda")
ximg, _ = ops.f16c_activation_image(x)
wimg, wexp, _ = ops.f16c_weight_image(w)
C = N // 2
aux = torch.randn((B, r // 2, r // 2, C), generator=g).cuda()
kw = dict(epilogue=ops.EPI_SPADE, aux=aux, aux_shift=1, mean=torch.zeros(C, device="cuda"), std=torch.ones(C, device="cuda"),
          out_padded=True, out_mode=4)
for _ in range(10):
    ops.conv3x3_f16c(ctx, ximg, wimg, wexp, bias, r, **kw)
torch.cuda.synchronize()
