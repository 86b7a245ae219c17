"""CPU: per-op known-answer tests of the generator oracle + golden regression."""
import os

import numpy as np
import pytest
import torch

from moonsuperresolution_amd import make_latent_noise, make_weights, synthetic_patches
from oracle import generator_ref as G
from tests.helpers import rel_linf

GOLD = os.path.join(os.path.dirname(__file__), "golden")
T64 = torch.float64


def test_same_padding_asymmetry():
    assert G.same_padding(8, 3, 1) == (1, 1)
    assert G.same_padding(8, 3, 2) == (0, 1)      # k3 s2 on even input: pad 0 before / 1 after
    assert G.same_padding(8, 4, 1) == (1, 2)      # k4 s1: 1 before / 2 after
    assert G.same_padding(8, 4, 2) == (1, 1)


def test_conv_same_delta_k4_shows_pad_before_one():
    # a delta at (0,0) convolved (cross-correlated) with a k=4 kernel: out[y,x] = k[0-y+1, 0-x+1]
    x = torch.zeros(1, 6, 6, 1, dtype=T64)
    x[0, 0, 0, 0] = 1
    k = torch.arange(16, dtype=T64).reshape(4, 4, 1, 1)
    y = G.conv2d_same(x, k)[0, :, :, 0]
    assert y[0, 0] == k[1, 1, 0, 0] and y[1, 1] == k[0, 0, 0, 0] and y[2, 2] == 0
    # delta at the far corner sees taps up to index 3-? : out[5,5] = k[1,1], out[3,3] = k[3,3]
    x = torch.zeros(1, 6, 6, 1, dtype=T64)
    x[0, 5, 5, 0] = 1
    y = G.conv2d_same(x, k)[0, :, :, 0]
    assert y[5, 5] == k[1, 1, 0, 0] and y[3, 3] == k[3, 3, 0, 0]


def test_conv_same_stride2_pad_after():
    x = torch.zeros(1, 4, 4, 1, dtype=T64)
    x[0, 3, 3, 0] = 1
    k = torch.arange(9, dtype=T64).reshape(3, 3, 1, 1)
    y = G.conv2d_same(x, k, stride=2)[0, :, :, 0]
    # out[1,1] reads in[2..4, 2..4] (row 4 is padding): the delta at (3,3) meets tap (1,1)
    assert y.shape == (2, 2) and y[1, 1] == k[1, 1, 0, 0] and y[0, 0] == 0


def test_resize_nearest_is_half_pixel():
    src = torch.arange(16, dtype=T64).reshape(1, 4, 4, 1)
    r = G.resize_nearest_halfpixel(src, 2)[0, :, :, 0]
    assert r.tolist() == [[5.0, 7.0], [13.0, 15.0]]            # picks indices 1 and 3, not 0 and 2
    assert torch.equal(G.resize_nearest_halfpixel(src, 4), src)


def test_upsample2x_and_moment_invariance():
    x = torch.arange(8, dtype=T64).reshape(1, 2, 2, 2)
    u = G.upsample2x(x)
    assert u.shape == (1, 4, 4, 2) and torch.equal(u[0, 1, 1], x[0, 0, 0]) and torch.equal(u[0, 2, 3], x[0, 1, 1])
    assert torch.allclose(u.mean((0, 1, 2)), x.mean((0, 1, 2))) and torch.allclose(u.var((0, 1, 2), unbiased=False), x.var((0, 1, 2), unbiased=False))


def test_spade_uses_batch_statistics_and_plain_gamma():
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((3, 4, 4, 8)))
    src = torch.from_numpy(rng.uniform(-0.5, 0.5, (3, 16, 16, 2)))
    w = {"p.conv.kernel": torch.zeros(3, 3, 2, 128, dtype=T64), "p.conv.bias": torch.ones(128, dtype=T64),
         "p.conv_gamma.kernel": torch.zeros(3, 3, 128, 8, dtype=T64), "p.conv_gamma.bias": torch.full((8,), 2.0, dtype=T64),
         "p.conv_beta.kernel": torch.zeros(3, 3, 128, 8, dtype=T64), "p.conv_beta.bias": torch.full((8,), 0.5, dtype=T64)}
    y = G.spade(x, src, w, "p")
    xn = (x - x.mean((0, 1, 2))) / torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-5)
    assert torch.allclose(y, 2.0 * xn + 0.5)            # gamma * x_hat + beta, NOT (1 + gamma)
    # coupling across the batch: changing sample 2 changes the output of sample 0
    x2 = x.clone()
    x2[2] += 5.0
    assert not torch.allclose(G.spade(x2, src, w, "p")[0], y[0])


def test_instance_norm_is_per_sample():
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((2, 5, 5, 3)))
    y = G.instance_norm(x, torch.ones(3, dtype=T64), torch.zeros(3, dtype=T64))
    assert torch.allclose(y.mean((1, 2)), torch.zeros(2, 3, dtype=T64), atol=1e-12)
    v = x.var((1, 2), unbiased=False)
    assert torch.allclose(y.var((1, 2), unbiased=False), v / (v + 1e-3))


def test_conv_transpose_matches_scatter_definition():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, 3, 3, 2))
    k = rng.standard_normal((4, 4, 5, 2))   # [kh, kw, Cout, Cin]
    y = G.conv2d_transpose_same_s2(torch.from_numpy(x), torch.from_numpy(k)).numpy()
    ref = np.zeros((1, 6, 6, 5))
    for iy in range(3):
        for ix in range(3):
            for kh in range(4):
                for kw in range(4):
                    oy, ox = 2 * iy - 1 + kh, 2 * ix - 1 + kw
                    if 0 <= oy < 6 and 0 <= ox < 6:
                        ref[0, oy, ox] += k[kh, kw] @ x[0, iy, ix]
    assert np.allclose(y, ref)


@pytest.mark.parametrize("variant", ["gaugan", "gaugan_no_kl"])
def test_spade_oracle_reproduces_golden(variant):
    g = np.load(os.path.join(GOLD, f"spade64_{variant}.npz"))
    w = make_weights(variant, 64, seed=1234, bias_scale=0.05)
    y = G.spade_call(synthetic_patches(2, 64, 0), w, variant, make_latent_noise(2, 256, 7), dtype=T64)
    assert y.shape == (2, 64, 64, 1)
    assert rel_linf(y, g["output"]) < 1e-9
    # float32 evaluation of the same oracle stays within the parity tolerance of the float64 one
    y32 = G.spade_call(synthetic_patches(2, 64, 0), w, variant, make_latent_noise(2, 256, 7), dtype=torch.float32)
    assert rel_linf(y32, g["output"]) < 1e-3


def test_variants_differ_only_in_latent():
    w = make_weights("cnn", 64, seed=1234, bias_scale=0.05)
    x = synthetic_patches(2, 64, 0)
    a = G.spade_call(x, w, "cnn", None, dtype=T64)
    b = G.spade_call(x, w, "gaugan_no_kl", None, dtype=T64)
    assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        G.spade_call(x, w, "gaugan", None)


def test_pix2pix_oracle_shapes_and_golden():
    g = np.load(os.path.join(GOLD, "pix2pix256.npz"))
    w = make_weights("pix2pix", 256, seed=1234, bias_scale=0.05)
    y = G.pix2pix_call(synthetic_patches(1, 256, 3), w, dtype=T64)
    assert y.shape == (1, 256, 256, 1) and np.abs(y).max() < 1.0     # tanh head
    assert abs(np.abs(y).mean() - g["checksum"]) < 1e-12
    assert np.allclose(y[0, 16::32, 16::32, 0], g["probe"], atol=1e-12)


# ---- more hand-computed known answers: every constant and ordering the restatement commits to -------------------
def test_constants_are_the_documented_defaults():
    """tfa InstanceNormalization epsilon 1e-3, Keras BatchNormalization epsilon 1e-3, Keras LeakyReLU() alpha 0.3 (pix2pix),
    alpha 0.2 where the reference passes it (blocks.py:30-34,65; networks.py:55), SPADE epsilon 1e-5 (spade.py:6)."""
    assert (G.IN_EPS, G.BN_EPS, G.P2P_LEAK, G.LEAK, G.SPADE_EPS) == (1e-3, 1e-3, 0.3, 0.2, 1e-5)
    x = torch.tensor([-2.0, 0.0, 3.0], dtype=T64)
    assert torch.equal(G.leaky_relu(x, 0.2), torch.tensor([-0.4, 0.0, 3.0], dtype=T64))
    assert torch.equal(G.leaky_relu(x, 0.3), torch.tensor([-0.6, 0.0, 3.0], dtype=T64))


def test_instance_norm_known_values():
    """Two pixels 1 and 3 of one sample / channel: mean 2, BIASED variance 1 -> (x - 2) / sqrt(1 + 1e-3), then gamma, beta."""
    x = torch.tensor([1.0, 3.0], dtype=T64).reshape(1, 1, 2, 1)
    y = G.instance_norm(x, torch.tensor([2.0], dtype=T64), torch.tensor([0.5], dtype=T64))
    want = np.array([-1.0, 1.0]) / np.sqrt(1.001) * 2.0 + 0.5
    assert np.allclose(y.numpy().ravel(), want, rtol=0, atol=1e-15)


def test_batch_norm_inference_known_values():
    """Keras BatchNormalization(training=False): (x - moving_mean) / sqrt(moving_var + 1e-3) * gamma + beta."""
    w = {"b.moving_mean": torch.tensor([1.0], dtype=T64), "b.moving_variance": torch.tensor([4.0], dtype=T64),
         "b.gamma": torch.tensor([2.0], dtype=T64), "b.beta": torch.tensor([-1.0], dtype=T64)}
    y = G.batch_norm_inference(torch.tensor([3.0], dtype=T64).reshape(1, 1, 1, 1), w, "b")
    assert abs(float(y) - ((3.0 - 1.0) / np.sqrt(4.001) * 2.0 - 1.0)) < 1e-15


def test_spade_known_values():
    """x = {0, 2} over the batch axis: batch mean 1, biased variance 1; zero conv kernels leave gamma = 3, beta = 0.25:
    out = 3 * (x - 1) / sqrt(1 + 1e-5) + 0.25 (spade.py:21-24: moments over (N,H,W), division by sqrt(var + eps))."""
    x = torch.tensor([0.0, 2.0], dtype=T64).reshape(2, 1, 1, 1)
    src = torch.zeros((2, 4, 4, 2), dtype=T64)
    w = {"p.conv.kernel": torch.zeros(3, 3, 2, 128, dtype=T64), "p.conv.bias": torch.zeros(128, dtype=T64),
         "p.conv_gamma.kernel": torch.zeros(3, 3, 128, 1, dtype=T64), "p.conv_gamma.bias": torch.tensor([3.0], dtype=T64),
         "p.conv_beta.kernel": torch.zeros(3, 3, 128, 1, dtype=T64), "p.conv_beta.bias": torch.tensor([0.25], dtype=T64)}
    y = G.spade(x, src, w, "p").numpy().ravel()
    assert np.allclose(y, np.array([-3.0, 3.0]) / np.sqrt(1.00001) + 0.25, rtol=0, atol=1e-15)


def test_mask_embedding_sees_the_half_pixel_resized_source_through_relu():
    """spade.py:17-18: h = relu(conv3x3(resize_nearest(source))).  A kernel that copies channel 1 of the centre tap makes
    gamma the resized DEM channel itself (negative values clipped by the ReLU): 4 -> 2 picks source pixels 1 and 3."""
    src = torch.zeros((1, 4, 4, 2), dtype=T64)
    src[0, :, :, 1] = torch.tensor([[-1.0, 2.0, -3.0, 4.0]] * 4, dtype=T64)
    k = torch.zeros(3, 3, 2, 128, dtype=T64)
    k[1, 1, 1, 0] = 1.0                                          # hidden channel 0 = DEM channel at the centre tap
    kg = torch.zeros(3, 3, 128, 1, dtype=T64)
    kg[1, 1, 0, 0] = 1.0                                         # gamma = hidden channel 0
    w = {"p.conv.kernel": k, "p.conv.bias": torch.zeros(128, dtype=T64), "p.conv_gamma.kernel": kg,
         "p.conv_gamma.bias": torch.zeros(1, dtype=T64), "p.conv_beta.kernel": torch.zeros(3, 3, 128, 1, dtype=T64),
         "p.conv_beta.bias": torch.zeros(1, dtype=T64)}
    x = torch.tensor([[1.0, -1.0], [-1.0, 1.0]], dtype=T64).reshape(1, 2, 2, 1)      # mean 0, variance 1
    y = G.spade(x, src, w, "p")[0, :, :, 0].numpy()
    gamma = np.array([[2.0, 4.0], [2.0, 4.0]])                   # source columns 1 and 3 (half-pixel nearest), relu keeps them
    assert np.allclose(y, gamma * np.array([[1.0, -1.0], [-1.0, 1.0]]) / np.sqrt(1.00001), atol=1e-15)


def test_dense_reshape_and_head_orders():
    """networks.py:41-42,55-56 with no residual blocks: Dense -> Reshape((sw, sw, 1024)) is NHWC order
    (index = (h * sw + w) * 1024 + c); the head is leaky_relu -> Conv2D(1, 4, 'same') with NO tanh (outputs beyond 1
    survive) and TF's even-kernel padding, 1 before / 2 after: tap (kh, kw) reads input (y + kh - 1, x + kw - 1)."""
    sw, hh, ww, c = 2, 1, 0, 5
    src = torch.zeros((1, 64 * sw, 64 * sw, 2), dtype=T64)
    dk = torch.zeros(3, sw * sw * 1024, dtype=T64)
    dk[0, (hh * sw + ww) * 1024 + c] = -7.0                      # latent e0 -> x0[h=1, w=0, c=5] = -7
    hk = torch.zeros(4, 4, 1024, 1, dtype=T64)
    hk[2, 1, c, 0] = 10.0                                        # reads (y + 1, x): out[0, 0] sees x0[1, 0]
    hk[3, 3, c, 0] = 100.0                                       # reads (y + 2, x + 2): always in the padding of a 2 x 2 map
    hk[0, 0, c, 0] = 1000.0                                      # reads (y - 1, x - 1): x0[1, 0] would land on out[2, 1] — outside
    w = {"gen.dense.kernel": dk, "gen.dense.bias": torch.zeros(sw * sw * 1024, dtype=T64), "gen.head.kernel": hk,
         "gen.head.bias": torch.tensor([0.5], dtype=T64)}
    y = G.generator(torch.tensor([[1.0, 0.0, 0.0]], dtype=T64), src, w)[0, :, :, 0].numpy()
    want = np.full((sw, sw), 0.5)
    want[0, 0] += 10.0 * (0.2 * -7.0)                            # leaky_relu(0.2) of -7, no tanh: -13.5
    assert np.allclose(y, want, atol=1e-15) and np.abs(y).max() > 1.0


def test_encoder_flatten_is_nhwc_and_first_block_has_no_norm():
    """networks.py:16-18,31-33: block 1 = strided conv + LeakyReLU(0.2) without InstanceNorm; Flatten on NHWC is
    index (h * W + w) * C + c; mean and variance are two Dense layers on the same flat vector."""
    src = torch.zeros((1, 4, 4, 2), dtype=T64)
    src[0, 2, 0, 1] = -5.0                                       # one pixel of the DEM channel
    k = torch.zeros(3, 3, 2, 3, dtype=T64)
    k[0, 0, 1, 2] = 1.0                                          # out channel 2 = DEM at tap (0,0): stride 2 pads 0 before
    hgt, wid, ch = 1, 0, 2                                       # source (2,0) -> output (1,0)
    flat_index = (hgt * 2 + wid) * 3 + ch
    mk = torch.zeros(12, 2, dtype=T64)
    mk[flat_index, 0] = 1.0
    vk = torch.zeros(12, 2, dtype=T64)
    vk[flat_index, 1] = 2.0
    w = {"enc.ds1.kernel": k, "enc.mean.kernel": mk, "enc.mean.bias": torch.tensor([0.0, 1.0], dtype=T64),
         "enc.variance.kernel": vk, "enc.variance.bias": torch.zeros(2, dtype=T64)}
    mean, var = G.encoder(src, w)
    assert torch.allclose(mean, torch.tensor([[0.2 * -5.0, 1.0]], dtype=T64))       # leaky_relu(0.2), no normalisation
    assert torch.allclose(var, torch.tensor([[0.0, 2.0 * 0.2 * -5.0]], dtype=T64))


def test_sampler_and_no_kl_latents():
    """sampling.py:16: z = mean + exp(0.5 * variance) * eps; GauGAN_no_KL / CNNSpade: z = mean + variance (model.py:267)."""
    w = make_weights("gaugan", 64, seed=5, bias_scale=0.05)
    x = synthetic_patches(2, 64, 4)
    eps = make_latent_noise(2, 256, 3)
    cap = {}
    G.spade_call(x, w, "gaugan", eps, dtype=T64, capture=cap)
    assert np.allclose(cap["z"], cap["enc.mean"] + np.exp(0.5 * cap["enc.variance"]) * eps, rtol=0, atol=1e-12)
    cap2 = {}
    G.spade_call(x, w, "gaugan_no_kl", dtype=T64, capture=cap2)
    assert np.allclose(cap2["z"], cap2["enc.mean"] + cap2["enc.variance"], rtol=0, atol=1e-12)


def test_residual_block_skip_is_identity_or_learned_3x3():
    """blocks.py:23-26,33-38: with equal filter counts the skip is x itself; otherwise conv_3(lrelu(spade_3(x))) — a 3x3
    conv WITH bias, not 1x1.  Zero main-path convs isolate the skip."""
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((1, 4, 4, 2)))
    src = torch.zeros((1, 8, 8, 2), dtype=T64)

    def spade_w(prefix, C):
        return {f"{prefix}.conv.kernel": torch.zeros(3, 3, 2, 128, dtype=T64), f"{prefix}.conv.bias": torch.zeros(128, dtype=T64),
                f"{prefix}.conv_gamma.kernel": torch.zeros(3, 3, 128, C, dtype=T64), f"{prefix}.conv_gamma.bias": torch.ones(C, dtype=T64),
                f"{prefix}.conv_beta.kernel": torch.zeros(3, 3, 128, C, dtype=T64), f"{prefix}.conv_beta.bias": torch.zeros(C, dtype=T64)}
    w = {**spade_w("b.spade_1", 2), **spade_w("b.spade_2", 2),
         "b.conv_1.kernel": torch.zeros(3, 3, 2, 2, dtype=T64), "b.conv_1.bias": torch.zeros(2, dtype=T64),
         "b.conv_2.kernel": torch.zeros(3, 3, 2, 2, dtype=T64), "b.conv_2.bias": torch.zeros(2, dtype=T64)}
    assert torch.allclose(G.residual_block(x, src, w, "b"), x)                     # identity skip
    k3 = torch.zeros(3, 3, 2, 3, dtype=T64)
    k3[1, 2, 0, 1] = 1.0                                                            # out channel 1 = input channel 0 one pixel to the right
    w3 = {**w, **spade_w("b.spade_3", 2), "b.conv_2.kernel": torch.zeros(3, 3, 2, 3, dtype=T64),
          "b.conv_2.bias": torch.zeros(3, dtype=T64), "b.conv_3.kernel": k3, "b.conv_3.bias": torch.tensor([0.0, 0.0, 7.0], dtype=T64)}
    w3["b.spade_2.conv_gamma.kernel"] = torch.zeros(3, 3, 128, 2, dtype=T64)
    y = G.residual_block(x, src, w3, "b")
    xn = (x - x.mean((0, 1, 2))) / torch.sqrt(x.var((0, 1, 2), unbiased=False) + 1e-5)
    a = G.leaky_relu(xn, 0.2)
    want1 = torch.zeros(4, 4, dtype=T64)
    want1[:, :-1] = a[0, :, 1:, 0]
    assert torch.allclose(y[0, :, :, 1], want1) and torch.allclose(y[0, :, :, 2], torch.full((4, 4), 7.0, dtype=T64))
    assert torch.allclose(y[0, :, :, 0], torch.zeros(4, 4, dtype=T64))
