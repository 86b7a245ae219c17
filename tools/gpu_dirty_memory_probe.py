"""Diagnostic (not a pytest): does a generator call depend on what its workspace held before?  The library allocates with
hipMalloc; after other work freed device memory the pages come back dirty, and a buffer that is read before it is
written (or assumed zero without a memset) shows up as a run-to-run difference.   usage: python tools/gpu_dirty_memory_probe.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import Generator, make_latent_noise, make_weights, synthetic_patches

for S, B in ((64, 2), (128, 3), (256, 4)):
    w = make_weights("gaugan", S, seed=1234, bias_scale=0.05)
    eps = make_latent_noise(B, 256, 7)
    x = torch.from_numpy(synthetic_patches(B, S, 5)).cuda()
    outs = []
    for fill in (None, 1e30, float("nan"), -3.0):
        if fill is not None:
            junk = torch.full((6 * 1024 ** 3 // 4,), fill, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            del junk
            torch.cuda.empty_cache()
        gen = Generator(S, B, variant="gaugan", weights=w, eps=eps)
        y = gen.forward_device(x).clone()
        y2 = gen.forward_device(x).clone()
        torch.cuda.synchronize()
        outs.append(y.cpu())
        print(f"S={S} B={B} fill={fill}: finite={bool(torch.isfinite(y).all())} repeat_equal={bool(torch.equal(y, y2))} "
              f"equal_to_first={bool(torch.equal(outs[0], y.cpu()))} max|diff|={float((outs[0] - y.cpu()).abs().max()):.3e}")
        gen.close()
