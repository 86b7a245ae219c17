"""Ad-hoc GPU bring-up script (not a pytest): per-block comparison of the HIP SPADE path with the oracle."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import Generator, make_weights, make_latent_noise, synthetic_patches
from oracle import generator_ref

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
variant = sys.argv[3] if len(sys.argv) > 3 else "gaugan"
w = make_weights(variant, S, seed=1234, bias_scale=0.05)
eps = make_latent_noise(B, 256, 7)
x = synthetic_patches(B, S, 0)
t = time.time(); gen = Generator(S, B, variant=variant, weights=w, eps=eps); print("create+load %.1fs" % (time.time() - t))
y = gen(x)
cap = {}
t = time.time(); ref = generator_ref.spade_call(x, w, variant, eps, dtype=torch.float64, capture=cap); print("oracle %.1fs" % (time.time() - t))

def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

sw = S // 64
print("mv   ", rel(gen.debug_tensor("ws.enc.mv", (B, 512)), np.concatenate([cap["enc.mean"], cap["enc.variance"]], 1)))
print("z    ", rel(gen.last_latent(), cap["z"]))
print("x0   ", rel(gen.debug_tensor("ws.gen.x0", (B, sw, sw, 1024)), cap["gen.x0"]))
f = [1024, 1024, 1024, 512, 256, 128]
for i in range(1, 7):
    r = sw << (i - 1)
    print(f"rb{i}.x1 ", rel(gen.debug_tensor(f"ws.gen.rb{i}.x1", (B, r, r, f[i-1])), cap[f"gen.rb{i}.x1"]))
    print(f"rb{i}.out", rel(gen.debug_tensor(f"ws.gen.rb{i}.out", (B, r, r, f[i-1])), cap[f"gen.rb{i}.out"]))
print("final", rel(y, ref), "max|ref|", np.abs(ref).max(), "finite", np.isfinite(y).all())
print("flops per call (lib) %.4f G" % (gen.forward_flops() / 1e9), "device MB", gen.device_bytes() / 2**20)
