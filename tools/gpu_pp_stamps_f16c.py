"""Diagnostic (not a pytest): per-step segment durations of the f16c form of the ping-pong conv (stamped build, see
tools/gpu_pp_stamps.py for the build recipe)."""
import sys
import torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import ops
ctx = ops.OpContext()
B, r, cin, N = 16, 32, 1024, 512
x = torch.randn((B, r + 2, r + 2, cin), device="cuda")
w = torch.randn((9, N, cin), device="cuda") * 0.01
bias = torch.zeros(N, device="cuda")
ximg, _ = ops.f16c_activation_image(x)
wimg, wexp, _ = ops.f16c_weight_image(w)
for _ in range(30):      # warm clocks
    ops.conv3x3_f16c(ctx, ximg, wimg, wexp, bias, r)
torch.cuda.synchronize()
