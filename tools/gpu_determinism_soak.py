"""Soak (not a pytest): the same batch through the same handle N times must give bit-identical outputs — a data race in
the persistent ping-pong kernel (LDS hazards, cross-tile staging) would show up as a rare mismatch."""
import sys, torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import Generator, make_weights, make_latent_noise, synthetic_patches
S, B, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
gen = Generator(S, B, weights=make_weights("gaugan", S, seed=1234), eps=make_latent_noise(B, 256, 7))
x = torch.from_numpy(synthetic_patches(B, S, seed=3)).cuda()
ref = gen.forward_device(x).clone()
bad = 0
out = torch.empty_like(ref)
for i in range(N):
    gen.forward_device(x, out=out)
    if not torch.equal(out, ref):
        bad += 1
        print("mismatch at iteration", i, float((out - ref).abs().max()), flush=True)
print(f"S={S} B={B}: {N} repeats, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
