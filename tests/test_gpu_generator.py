"""GPU (-m gpu): the HIP generator(call) through the C ABI against the CPU oracle and the golden vectors.

Tolerance: BASELINE.json north_star states <= 1e-3 relative L-infinity vs the reference generator output in fp32;
every check here uses rel_linf = max|y - ref| / max|ref| against the float64 oracle and requires <= 1e-3
(observed 2-5e-5 with the default conv arithmetic — "f16c", which at the small shapes of this file selects the split-bf16
kernels throughout — that every test here runs unless it names a precision; the
exact-fp32 MFMA mode and the full-size BASELINE configurations are covered by tests/test_gpu_baseline_configs.py)."""
import os

import numpy as np
import pytest
import torch

from moonsuperresolution_amd import make_latent_noise, make_weights, synthetic_patches
from tests.helpers import rel_linf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-3


@pytest.fixture(scope="module")
def Generator(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from moonsuperresolution_amd import Generator
    return Generator


@pytest.mark.parametrize("variant", ["gaugan", "gaugan_no_kl"])
def test_spade64_matches_golden_and_blocks(Generator, variant):
    g = np.load(os.path.join(GOLD, f"spade64_{variant}.npz"))
    w = make_weights(variant, 64, seed=1234, bias_scale=0.05)
    gen = Generator(64, 2, variant=variant, weights=w, eps=make_latent_noise(2, 256, 7))
    y = gen(synthetic_patches(2, 64, 0), training=False)
    assert y.shape == (2, 64, 64, 1) and y.dtype == np.float32
    assert rel_linf(y, g["output"]) <= TOL
    # per-block checksums of the fp64 oracle (mean |x| of every ResidualBlock output)
    f = [1024, 1024, 1024, 512, 256, 128]
    for i in range(1, 7):
        r = 1 << (i - 1)
        out = gen.debug_tensor(f"ws.gen.rb{i}.out", (2, r, r, f[i - 1]))
        assert abs(np.abs(out).mean() / g[f"gen_rb{i}_out"] - 1) <= TOL, i
    gen.close()


def test_profile_modes_and_clone(Generator):
    """msr_profile_enable: mode 1 brackets every launch, mode 2 only runs of conv launches (bench.py's timed region);
    both count the same conv launches and FLOPs, and msr_profile_runs returns ordered intervals.  A clone is a second
    handle with the same weights: bit-identical output."""
    w = make_weights("gaugan", 64, seed=1234, bias_scale=0.05)
    eps = make_latent_noise(2, 256, 7)
    gen = Generator(64, 2, variant="gaugan", weights=w, eps=eps)
    x = torch.from_numpy(synthetic_patches(2, 64, 0)).cuda()
    y0 = gen.forward_device(x).cpu().numpy()
    stats = {}
    for mode in (1, 2):
        gen.profile(mode)
        ref = torch.cuda.Event(enable_timing=True)
        ref.record()
        for _ in range(3):
            gen.forward_device(x)
        stats[mode] = gen.profile_read()
        runs = gen.profile_runs(ref)
        gen.profile(0)
        conv = stats[mode]["conv_igemm_bf16x3"]
        assert sum(r[3] for r in runs) == conv["launches"] and abs(sum(r[2] for r in runs) - conv["flops"]) < 1e-3 * conv["flops"]
        assert all(0 <= a <= b for a, b, _, _ in runs)
    assert stats[1]["conv_igemm_bf16x3"]["launches"] == stats[2]["conv_igemm_bf16x3"]["launches"] == 3 * 34
    assert stats[1]["conv_igemm_bf16x3"]["flops"] == stats[2]["conv_igemm_bf16x3"]["flops"]
    assert set(stats[2]) == {"conv_igemm_bf16x3"} and len(stats[1]) > 4
    with pytest.raises(ValueError):
        gen.profile(3)
    twin = gen.clone()
    assert np.array_equal(twin.forward_device(x).cpu().numpy(), y0)
    twin.close()
    gen.close()


def test_graph_replay_equals_eager(Generator):
    """msr_graph_enable: the launch plan (two streams, fork / joins) captured per buffer triple and replayed with one
    hipGraphLaunch gives bit-identical results, for repeated calls, several output buffers, a changed input content,
    and after a load() (the re-plan drops the graphs)."""
    w = make_weights("gaugan", 64, seed=1234, bias_scale=0.05)
    w2 = make_weights("gaugan", 64, seed=99, bias_scale=0.05)
    eps = make_latent_noise(2, 256, 7)
    gen = Generator(64, 2, variant="gaugan", weights=w, eps=eps)
    x = torch.from_numpy(synthetic_patches(2, 64, 0)).cuda()
    x_other = torch.from_numpy(synthetic_patches(2, 64, 5)).cuda()
    ref = gen.forward_device(x).clone()
    ref_other = gen.forward_device(x_other).clone()
    assert torch.equal(gen.forward_device(x_other), ref_other) and torch.equal(gen.forward_device(x), ref)   # eager is repeatable
    gen.use_graph(True)
    st = torch.cuda.Stream()
    outs = [torch.empty((2, 64, 64, 1), device="cuda") for _ in range(3)]
    with torch.cuda.stream(st):
        for o in outs + outs + outs:                 # first pass eager (first sighting), second captures, third replays
            gen.forward_device(x, out=o)
        st.synchronize()
        assert all(torch.equal(o, ref) for o in outs)
        # more repeating triples than the cache holds (8): the least recently used graph is evicted, results stay right
        many = [torch.empty((2, 64, 64, 1), device="cuda") for _ in range(10)]
        for o in many + many + many:
            gen.forward_device(x, out=o)
        st.synchronize()
        assert all(torch.equal(o, ref) for o in many)
        # one-shot triples (a fresh output per call) never evict the graphs of the repeating ones
        for _ in range(20):
            assert torch.equal(gen.forward_device(x, out=torch.empty((2, 64, 64, 1), device="cuda")), ref)
        xin = x.clone()
        gen.forward_device(xin, out=outs[0])
        gen.forward_device(xin, out=outs[0])         # second sighting: captured
        st.synchronize()
        assert torch.equal(outs[0], ref)
        xin.copy_(x_other)                           # same pointer, new content: the graph reads the buffer, not a copy
        gen.forward_device(xin, out=outs[0])
        st.synchronize()
        assert not torch.equal(outs[0], ref), "the replay did not see the new content of its input buffer (or did not run)"
        if not torch.equal(outs[0], ref_other):      # diagnostics: which side moved?
            gen.use_graph(False)
            again = gen.forward_device(x_other).clone()
            again_xin = gen.forward_device(xin).clone()
            st.synchronize()
            raise AssertionError(f"graph replay != eager reference: eager(x_other) now == reference: {torch.equal(again, ref_other)}, "
                                 f"eager(xin) now == reference: {torch.equal(again_xin, ref_other)}, eager(xin) == replay: "
                                 f"{torch.equal(again_xin, outs[0])}, xin == x_other: {torch.equal(xin, x_other)}, "
                                 f"max |replay - reference| = {float((outs[0] - ref_other).abs().max()):.3e}")
        gen.load(w2)
        y2 = gen.forward_device(x, out=outs[1]).clone()
        st.synchronize()
    gen.use_graph(False)
    assert torch.equal(gen.forward_device(x), y2) and not torch.equal(y2, ref)
    gen.close()


def test_forward_gated_equals_plain_forward(Generator):
    """msr_forward_gated (include/moonsr.h): two handles on two streams, each call's matrix-bound part gated on the event the
    previous call (on the other handle) recorded at its end — the software pipeline of the tile loop (MSR_TILER_GATED).
    Results must equal plain msr_forward bit for bit, whatever the interleaving."""
    w = make_weights("gaugan", 64, seed=1234, bias_scale=0.05)
    eps = make_latent_noise(2, 256, 7)
    g0 = Generator(64, 2, variant="gaugan", weights=w, eps=eps)
    g1 = g0.clone()
    xs = [torch.from_numpy(synthetic_patches(2, 64, seed)).cuda() for seed in range(6)]
    want = [g0.forward_device(x).clone() for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.empty((2, 64, 64, 1), device="cuda") for _ in xs]
    gate = None
    for i, x in enumerate(xs):
        g, st = (g0, g1)[i & 1], streams[i & 1]
        with torch.cuda.stream(st):
            g.forward_device(x, out=outs[i], gate=gate)
            gate = torch.cuda.Event()
            gate.record(st)
    torch.cuda.synchronize()
    for o, ref in zip(outs, want):
        assert torch.equal(o, ref)
    # a gate that has long fired, and no gate at all, on the same handle
    ev = torch.cuda.Event()
    ev.record()
    torch.cuda.synchronize()
    assert torch.equal(g1.forward_device(xs[0], gate=ev), want[0]) and torch.equal(g1.forward_device(xs[1]), want[1])
    g0.close(); g1.close()


def test_cnn_variant_equals_no_kl(Generator):
    w = make_weights("cnn", 64, seed=1234, bias_scale=0.05)
    x = synthetic_patches(2, 64, 0)
    a = Generator(64, 2, variant="cnn", weights=w)(x)
    b = Generator(64, 2, variant="gaugan_no_kl", weights=w)(x)
    assert np.array_equal(a, b)
    g = np.load(os.path.join(GOLD, "spade64_gaugan_no_kl.npz"))
    assert rel_linf(a, g["output"]) <= TOL


def test_zero_padding_patch_couples_through_batch_statistics(Generator):
    """process_full_tiles.py:468-474 pads the last batch with zero patches; SPADE's batch moments (spade.py:21)
    make the real patch's output depend on them — parity only holds per call with identical composition."""
    g0 = np.load(os.path.join(GOLD, "spade64_gaugan.npz"))
    gz = np.load(os.path.join(GOLD, "spade64_zero.npz"))
    w = make_weights("gaugan", 64, seed=1234, bias_scale=0.05)
    gen = Generator(64, 2, variant="gaugan", weights=w, eps=make_latent_noise(2, 256, 7))
    x = synthetic_patches(2, 64, 0)
    x[1] = 0.0
    y = gen(x)
    assert rel_linf(y, gz["output"]) <= TOL
    assert rel_linf(y[0], g0["output"][0]) > 10 * TOL      # same patch 0, different batch mate -> different result
    gen.close()


def test_float64_batch_and_list_input_accepted(Generator):
    """np.array(batch) is float64 when zero patches were appended (process_full_tiles.py:472): same result."""
    w = make_weights("gaugan_no_kl", 64, seed=1234)
    gen = Generator(64, 2, variant="gaugan_no_kl", weights=w)
    x = synthetic_patches(2, 64, 1)
    a = gen(x)
    b = gen([x[0].astype(np.float64), x[1].astype(np.float64)], training=False)
    assert np.array_equal(a, b)
    assert np.array_equal(np.array(a)[:, :, :, -1], a[..., 0])     # what processBatch does with the result
    gen.close()


def test_error_behaviour(Generator):
    w = make_weights("gaugan", 64, seed=1234)
    gen = Generator(64, 2, variant="gaugan", weights=w, eps=7)
    with pytest.raises(ValueError):
        gen(np.zeros((3, 64, 64, 2), np.float32))           # batch != batch_size (sampling.py:13-15)
    with pytest.raises(ValueError):
        gen(np.zeros((2, 32, 32, 2), np.float32))
    with pytest.raises(ValueError):
        gen(np.zeros((2, 64, 64, 2), np.float32), training=True)
    bad = dict(w)
    bad.pop("gen.rb3.conv_2.kernel")
    with pytest.raises(ValueError):
        Generator(64, 2, variant="gaugan", weights=bad)
    bad = dict(w)
    bad["gen.head.kernel"] = np.zeros((3, 3, 128, 1), np.float32)
    with pytest.raises(ValueError):
        Generator(64, 2, variant="gaugan", weights=bad)
    with pytest.raises(ValueError):
        Generator(96, 2)
    gen.close()


def test_unseeded_sampler_draws_fresh_noise(Generator):
    w = make_weights("gaugan", 64, seed=1234)
    gen = Generator(64, 2, variant="gaugan", weights=w, eps=None)
    x = synthetic_patches(2, 64, 2)
    assert not np.array_equal(gen(x), gen(x))               # tf.random.normal is unseeded (sampling.py:13)
    gen.close()


def test_spade256_against_oracle(Generator):
    """BASELINE config 2 geometry (S=256) at B=2 so the fp64 oracle finishes in seconds."""
    from oracle import generator_ref
    w = make_weights("gaugan", 256, seed=1234, bias_scale=0.05)
    eps = make_latent_noise(2, 256, 7)
    x = synthetic_patches(2, 256, 0)
    gen = Generator(256, 2, variant="gaugan", weights=w, eps=eps)
    y = gen(x)
    cap = {}
    ref = generator_ref.spade_call(x, w, "gaugan", eps, dtype=torch.float64, capture=cap)
    assert rel_linf(y, ref) <= TOL
    assert rel_linf(gen.last_latent(), cap["z"]) <= TOL
    assert rel_linf(gen.debug_tensor("ws.gen.rb6.x1", (2, 128, 128, 128)), cap["gen.rb6.x1"]) <= TOL
    assert abs(gen.forward_flops() / 2 / 1e9 - 175.652) < 1e-3
    gen.close()


def test_full_size_configs_properties(Generator):
    """BASELINE configs 2 and 3 at full batch: size-independent properties (determinism, batch-permutation
    equivariance, FLOP count).  The oracle comparison at these sizes is tests/test_gpu_baseline_configs.py."""
    for S, B in ((256, 16), (512, 8), (512, 12)):      # (512, 12) = the production setting of run_GAN.sh:24-26
        gen = Generator(S, B, variant="gaugan", weights=1234, eps=7)
        x = torch.from_numpy(synthetic_patches(B, S, 11)).cuda()
        a = gen.forward_device(x).clone()
        b = gen.forward_device(x).clone()
        assert torch.equal(a, b)                               # deterministic: no atomics on the path
        assert a.shape == (B, S, S, 1) and torch.isfinite(a).all()
        # permuting the batch permutes the output (batch statistics are permutation invariant); the latent
        # noise is per row, so permute it too
        perm = torch.arange(B - 1, -1, -1)
        eps = torch.from_numpy(make_latent_noise(B, 256, 7)).cuda()
        c = gen.forward_device(x[perm].contiguous(), eps=eps[perm].contiguous())
        err = float((c[perm] - a).abs().max() / a.abs().max())
        assert err <= 1e-4, err
        assert abs(gen.forward_flops() / B / 1e9 - (175.652 if S == 256 else 702.607)) < 1e-3
        gen.close()
        del gen
        torch.cuda.empty_cache()


def test_pix2pix_against_oracle(Generator):
    from oracle import generator_ref
    g = np.load(os.path.join(GOLD, "pix2pix256.npz"))
    w = make_weights("pix2pix", 256, seed=1234, bias_scale=0.05)
    x = synthetic_patches(1, 256, 3)
    gen = Generator(256, 1, variant="pix2pix", weights=w)
    y = gen(x)
    ref = generator_ref.pix2pix_call(x, w, dtype=torch.float64)
    assert y.shape == (1, 256, 256, 1) and np.abs(y).max() < 1.0
    assert rel_linf(y, ref) <= TOL
    assert np.allclose(y[0, 16::32, 16::32, 0], g["probe"], atol=TOL * g["absmax"])
    assert abs(gen.forward_flops() / 1e9 - 11.929) < 1e-3
    gen.close()


def test_pix2pix_batched_and_repeatable(Generator):
    """B = 3 (not a power of two: the low-resolution MFMA tiles span several images and mask the tail) and a second
    call on the same handle (the zero borders of the concat buffers must survive a forward)."""
    from oracle import generator_ref
    w = make_weights("pix2pix", 256, seed=77, bias_scale=0.05)
    x = synthetic_patches(3, 256, 5)
    gen = Generator(256, 3, variant="pix2pix", weights=w)
    y1 = gen(x)
    y2 = gen(x)
    ref = generator_ref.pix2pix_call(x, w, dtype=torch.float64)
    assert rel_linf(y1, ref) <= TOL
    assert np.array_equal(y1, y2)
    gen.close()


def test_bf16x3_precision_within_tolerance(Generator):
    """MSR_FLAG_BF16X3: conv products as 3-term split-bf16 on the bf16 MFMA; everything else fp32.
    Same <= 1e-3 bar against the float64 oracle (observed ~1e-4)."""
    from oracle import generator_ref
    g = np.load(os.path.join(GOLD, "spade64_gaugan.npz"))
    w = make_weights("gaugan", 64, seed=1234, bias_scale=0.05)
    gen = Generator(64, 2, variant="gaugan", weights=w, eps=make_latent_noise(2, 256, 7), precision="bf16x3")
    y = gen(synthetic_patches(2, 64, 0))
    assert rel_linf(y, g["output"]) <= TOL
    gen.close()
    w = make_weights("gaugan", 256, seed=1234, bias_scale=0.05)
    eps = make_latent_noise(2, 256, 7)
    x = synthetic_patches(2, 256, 0)
    gen = Generator(256, 2, variant="gaugan", weights=w, eps=eps, precision="bf16x3")
    y = gen(x)
    ref = generator_ref.spade_call(x, w, "gaugan", eps, dtype=torch.float64)
    err = rel_linf(y, ref)
    print("bf16x3 S=256 rel Linf", err)
    assert err <= TOL
    gen.close()
    with pytest.raises(ValueError):
        Generator(64, 2, precision="int4")              # not one of fp32 / bf16x3 / f16c / bf16x3_gbf16 / fp8


def test_load_GAN_model_from_savedmodel_dirs(Generator, tmp_path):
    """process_full_tiles.py:13-31: load_GAN_model(path, image_size, batch_size) with path+'generator' / path+'encoder'."""
    from moonsuperresolution_amd import load_GAN_model, tf_checkpoint
    from tests.test_tf_checkpoint import _keras_keys
    w = make_weights("gaugan", 64, seed=1234, bias_scale=0.05)
    gen_keys, enc_keys = _keras_keys(w)
    root = str(tmp_path) + "/"
    tf_checkpoint.write_tensor_bundle(root + "generator/variables/variables", gen_keys)
    tf_checkpoint.write_tensor_bundle(root + "encoder/variables/variables", enc_keys)
    gen = load_GAN_model(root, 64, 2, eps=make_latent_noise(2, 256, 7))
    y = gen(synthetic_patches(2, 64, 0))
    g = np.load(os.path.join(GOLD, "spade64_gaugan.npz"))
    assert rel_linf(y, g["output"]) <= TOL
    gen.close()
    with pytest.raises(AssertionError):
        load_GAN_model(root + "missing/", 64, 2)
