"""N generator calls at B = 1 on one handle / stream (per-layer table of the single-tile latency path).
usage: python profiles/run_forwards_b1.py <S> <N> [precision]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from moonsuperresolution_amd import Generator, make_latent_noise, make_weights, synthetic_patches

S, n = int(sys.argv[1]), int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 else "f16c"
gen = Generator(S, 1, variant="gaugan", weights=make_weights("gaugan", S, seed=1234), eps=make_latent_noise(1, 256, 7), precision=prec)
x = torch.from_numpy(synthetic_patches(1, S, seed=0)).cuda()
out = torch.empty((1, S, S, 1), device="cuda")
for _ in range(n):
    gen.forward_device(x, out=out)
torch.cuda.synchronize()
print(f"{n} calls of GauGAN({S},1) ({prec}) done")
