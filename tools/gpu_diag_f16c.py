import sys, numpy as np, torch
sys.path.insert(0, ".")
from moonsuperresolution_amd import ops
from tests.test_gpu_conv_kernel import ref_conv
from tests.helpers import rel_linf
ctx = ops.OpContext()
g = torch.Generator(device="cpu").manual_seed(1)
B, r, cin, cout = 2, 16, 64, 128
x = torch.randn((B, r, r, cin), generator=g).cuda()
w = (torch.randn((3, 3, cin, cout), generator=g) / np.sqrt(9 * cin)).cuda()
b = torch.zeros(cout).cuda()
ximg, (xh, x8, xl) = ops.f16c_activation_image(ops.pad_nhwc(x))
wimg, wexp, (wh, w8, wl) = ops.f16c_weight_image(ops.kernel_layout(w))
y = torch.from_numpy(ops.conv3x3_f16c(ctx, ximg, wimg, wexp, b, r).cpu().numpy()).double()
hw = lambda t: t.cpu().permute(0, 2, 1).reshape(3, 3, cin, cout)
z = torch.zeros(cout, dtype=torch.float64)
c = lambda a, k: ref_conv(a[:, 1:-1, 1:-1].cpu(), hw(k), z, 1)
main, t_even, t_odd = c(xh, wh), c(x8, wl), c(xl, w8)
full = ref_conv(x, w, b, 1)
print("wexp sample", [hex(int(v)) for v in wexp[:4]])
for name, ref in (("main", main), ("main+even", main + t_even), ("main+odd", main + t_odd), ("main+both", main + t_even + t_odd),
                  ("main+2*even", main + 2 * t_even), ("main+even*2^?", None)):
    if ref is None: continue
    print(f"{name:14s} {rel_linf(y.numpy(), ref.numpy()):.3e}")
res = (y - main)
# least squares fit of residual onto t_even, t_odd
A = torch.stack([t_even.flatten(), t_odd.flatten()], 1)
sol = torch.linalg.lstsq(A, res.flatten().unsqueeze(1)).solution.flatten()
print("residual = %.4f * even + %.4f * odd ; |res|max %.3e |even|max %.3e |odd|max %.3e" % (sol[0], sol[1], res.abs().max(), t_even.abs().max(), t_odd.abs().max()))
print("vs full fp64 conv", rel_linf(y.numpy(), full.numpy()))
