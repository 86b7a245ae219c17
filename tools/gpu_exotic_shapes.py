"""Diagnostic (not a pytest): oracle parity at shapes outside the BASELINE set (other tile counts, K ranges, odd batch)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import Generator, make_latent_noise, make_weights, synthetic_patches
from oracle import generator_ref
torch.set_num_threads(16)
for S, B in ((1024, 2), (128, 16), (256, 5), (512, 3), (64, 16)):
    w = make_weights("gaugan", S, seed=1234, bias_scale=0.05)
    eps = make_latent_noise(B, 256, 7)
    x = synthetic_patches(B, S, 0)
    t0 = time.time()
    ref = np.asarray(generator_ref.spade_call(x, w, "gaugan", eps, dtype=torch.float32), np.float64)
    t1 = time.time()
    for prec in ("f16c", "f16", "bf16x3"):
        gen = Generator(S, B, variant="gaugan", weights=w, eps=eps, precision=prec)
        y = gen(x, training=False)
        gen.close()
        print(f"GauGAN({S},{B}) {prec}: rel L-inf vs fp32 oracle {np.abs(y - ref).max() / np.abs(ref).max():.3e}  (oracle {t1 - t0:.1f} s)", flush=True)
