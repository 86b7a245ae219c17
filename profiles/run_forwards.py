"""Exactly N generator calls on ONE handle / stream (for rocprofv3 passes: per-layer tables and PMC counters need a
known call count and launches in plan order, which bench.py's two-stream pipeline does not give).
usage: python profiles/run_forwards.py <spade256|spade512> <N> [fp32|bf16x3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from moonsuperresolution_amd import Generator, make_latent_noise, make_weights, synthetic_patches

wl, n = sys.argv[1], int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 else "f16c"
S, B = (256, 16) if wl == "spade256" else (512, 8)
gen = Generator(S, B, variant="gaugan", weights=make_weights("gaugan", S, seed=1234), eps=make_latent_noise(B, 256, 7),
                precision=prec)
x = torch.from_numpy(synthetic_patches(B, S, seed=0)).cuda()
out = torch.empty((B, S, S, 1), device="cuda")
for _ in range(n):
    gen.forward_device(x, out=out)
torch.cuda.synchronize()
print(f"{n} calls of {wl} ({prec}) done")
