"""Diagnostic (not a pytest): in-kernel s_memtime stamps of conv_gb_resident (csrc/conv_gbr.hip): per work item the cycles of
phase 1 (patch + embedding halo into LDS) and of the sweep (all channel blocks incl. their epilogues).

    mkdir -p /tmp/st/pkg && cp -r moonsuperresolution_amd/csrc /tmp/st/pkg/csrc && cp -r include /tmp/st/include
    make -C /tmp/st/pkg/csrc clean all EXTRA="-DMSR_DIAG_BUILD -DMSR_GB_STAMPS"
    MSR_ALLOW_DIAG_BUILD=1 MSR_LIB=/tmp/st/pkg/csrc/libmoonsr_hip.so python tools/gpu_gbr_stamps.py [r] [C]
Stamps of workgroup 8, per wave: item start | after phase 1 | item end (repeated per item): deltas alternate phase 1, sweep."""
import sys
import torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import ops
ctx = ops.OpContext()
r = int(sys.argv[1]) if len(sys.argv) > 1 else 256
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B, S = 8, 512
g = torch.Generator(device="cpu").manual_seed(1)
src = (torch.rand((B, S, S, 2), generator=g) - 0.5).cuda()
we = (torch.randn((3, 3, 2, 128), generator=g) / 3).cuda()
be = (0.1 * torch.randn(128, generator=g)).cuda()
wg = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
wb = (torch.randn((3, 3, 128, C), generator=g) / 34).cuda()
bg, bb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
x = (3 + 2 * torch.randn((B, r // 2, r // 2, C), generator=g)).cuda()
mean = x.mean((0, 1, 2)).contiguous()
std = torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-5).contiguous()
w, bias = ops.spade_layout(wg, wb, bg, bb)
wimg = ops.gbr_weight_image(w)
for _ in range(10):      # warm clocks; every launch prints its stamps: read the last block
    ops.spade_gbr(ctx, src, we, be, wimg, bias, r, x, 1, mean, std)
torch.cuda.synchronize()
