"""Diagnostic (not a pytest): in-kernel s_memtime stamps of the software-pipelined f16c conv (conv_sw.hip).

    cp -r moonsuperresolution_amd/csrc /tmp/csrc_stamps && cp -r include /tmp/include
    make -C /tmp/csrc_stamps clean all EXTRA="-DMSR_DIAG_BUILD -DMSR_SW_STAMPS=1"    # =2: one stamp per phase (E, O, C) instead of per tap pair
    MSR_ALLOW_DIAG_BUILD=1 MSR_LIB=/tmp/csrc_stamps/libmoonsr_hip.so python tools/gpu_sw_stamps.py [main|gb]
Ideal (MFMA cycles only): 2048 per tap pair = 512 (E) + 512 (O) + 1024 (C); 18432 per body of 18 K-steps."""
import sys
import torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import ops
ctx = ops.OpContext()
kind = sys.argv[1] if len(sys.argv) > 1 else "main"
B, r, cin, N = (16, 32, 1024, 512) if kind == "main" else (8, 64, 128, 1024)
x = torch.randn((B, r + 2, r + 2, cin), device="cuda")
w = torch.randn((9, N, cin), device="cuda") * 0.01
bias = torch.zeros(N, device="cuda")
ximg, _ = ops.f16c_activation_image(x)
wimg, wexp, _ = ops.f16c_weight_image(w)
for _ in range(20):      # warm clocks (every launch prints its stamps; read the last block of lines)
    ops.conv3x3_f16c(ctx, ximg, wimg, wexp, bias, r)
torch.cuda.synchronize()
