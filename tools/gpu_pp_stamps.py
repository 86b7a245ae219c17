"""Diagnostic (not a pytest): run one long-K ping-pong conv with a stamped build of the library, which prints the
per-step segment durations (R, barrier, M, barrier) of waves 0 (X) and 4 (Y) of one workgroup in shader-clock cycles.

    cp -r moonsuperresolution_amd/csrc /tmp/csrc_stamps && cp -r include /tmp/include   # keeps the product objects clean
    make -C /tmp/csrc_stamps clean all EXTRA="-DMSR_DIAG_BUILD -DMSR_PP_STAMPS"       # (the Makefile reads ../../include: copy the tree)
    MSR_ALLOW_DIAG_BUILD=1 MSR_LIB=/path/to/stamped/libmoonsr_hip.so python tools/gpu_pp_stamps.py
"""
import sys
import torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import ops
ctx = ops.OpContext()
B, r, cin, N = 16, 32, 1024, 512
x = torch.randn((B, r + 2, r + 2, cin), device="cuda")
w = torch.randn((9, N, cin), device="cuda") * 0.01
bias = torch.zeros(N, device="cuda")
out = torch.empty((B, r, r, N), device="cuda")
for _ in range(30):      # warm clocks
    ops.conv3x3(ctx, x, w, bias, r, tile=4, out=out, precision="bf16x3")
torch.cuda.synchronize()
ops.conv3x3(ctx, x, w, bias, r, tile=5, out=out, precision="bf16x3")
torch.cuda.synchronize()
