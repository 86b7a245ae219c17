"""GPU (-m gpu): oracle parity on the exact BASELINE.json configurations, in every conv arithmetic the library offers.

The conv kernel a layer runs on is chosen from its shape (conv_igemm.hip: conv_pick_tile / conv_pick_ksplit on
M = B * r^2; api.hip: pick_conv_variant), so the B = 2 end-to-end tests of test_gpu_generator.py run other kernels
for rb3-rb5 than bench.py does (split-K small tiles instead of the persistent ping-pong kernel with fused output
moments).  These tests run the bench shapes themselves — GauGAN(256, 16), GauGAN(512, 8) and the production
setting GauGAN(512, 12) of run_GAN.sh:24-26 — against the CPU restatement of GauGAN.call
(spade/models/model.py:564-567, spade.py:16-25; oracle/generator_ref.py), float64 at S = 256 and float32 at S = 512
(its own rounding, ~4e-6, is far below the bar).

Tolerance (BASELINE.json north_star): relative L-infinity max|y - ref| / max|ref| <= 1e-3 in every mode.  "f16c" (the
default: fp16 main term + fp8 cross terms), "bf16x3" and "fp32" sit 20-300x inside it (4e-5 / 2e-5 / 4e-6); the opt-in "bf16x3_gbf16" (2-term fp16 products in the gamma|beta convs:
the weight is rounded to one fp16, 2^-12 per product) is bounded by the same bar with a smaller margin (measured
2-5e-4).  The measured values are printed and appended to gpurun_out/parity_baseline_configs.jsonl.
"""
import json
import os

import numpy as np
import pytest
import torch

from moonsuperresolution_amd import make_latent_noise, make_weights, synthetic_patches
from tests.helpers import rel_linf

pytestmark = pytest.mark.gpu
TOL = 1e-3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN_FILTERS = [1024, 1024, 1024, 512, 256, 128]


class _Pick(dict):
    """capture dict that keeps only the named tensors (the oracle offers every block output; at S = 512 they are GBs)."""

    def __init__(self, wanted):
        super().__init__()
        self.wanted = set(wanted)

    def __setitem__(self, k, v):
        if k in self.wanted:
            super().__setitem__(k, v)


_ORACLE = {}


def _oracle(S, B):
    """(x, weights, eps, oracle output, captured tensors) of one configuration, computed once per session."""
    if (S, B) not in _ORACLE:
        from oracle import generator_ref
        _ORACLE.clear()                                    # one configuration's tensors at a time
        w = make_weights("gaugan", S, seed=1234, bias_scale=0.05)
        eps = make_latent_noise(B, 256, 7)
        x = synthetic_patches(B, S, 0)
        cap = _Pick(["z", "gen.rb5.x1", "gen.rb4.out"])
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        dtype = torch.float64 if S <= 256 else torch.float32
        ref = generator_ref.spade_call(x, w, "gaugan", eps, dtype=dtype, capture=cap)
        _ORACLE[(S, B)] = (x, w, eps, np.asarray(ref, np.float64), dict(cap), str(dtype).split(".")[-1])
    return _ORACLE[(S, B)]


def _record(**kw):
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_baseline_configs.jsonl"), "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass
    print("parity", kw)


@pytest.mark.parametrize("precision", ["bf16x3", "fp32", "bf16x3_gbf16", "f16c"])
@pytest.mark.parametrize("S,B", [(256, 16), (512, 8), (512, 12)])
def test_baseline_config_matches_oracle(hip_lib, S, B, precision):
    from moonsuperresolution_amd import Generator
    x, w, eps, ref, cap, oracle_dtype = _oracle(S, B)
    gen = Generator(S, B, variant="gaugan", weights=w, eps=eps, precision=precision)
    y = gen(x, training=False)
    err = rel_linf(y, ref)
    sw = S // 64
    # a tensor written by the persistent ping-pong kernel with fused output moments (rb5 conv_1: r = 16 sw, 256 ch),
    # the batch moments that kernel emitted (they feed spade_2), and a block output with the residual epilogue
    r5, r4 = sw * 16, sw * 8
    x1 = gen.debug_tensor("ws.gen.rb5.x1", (B, r5, r5, 256))
    e_x1 = rel_linf(x1, cap["gen.rb5.x1"])
    x1_ref = np.asarray(cap["gen.rb5.x1"], np.float64)
    m_ref = x1_ref.mean((0, 1, 2))
    s_ref = np.sqrt(x1_ref.var((0, 1, 2)) + 1e-5)
    e_mean = float(np.abs(gen.debug_tensor("ws.gen.rb5.mean1", (256,)) - m_ref).max() / s_ref.max())
    e_std = rel_linf(gen.debug_tensor("ws.gen.rb5.std1", (256,)), s_ref)
    e_rb4 = rel_linf(gen.debug_tensor("ws.gen.rb4.out", (B, r4, r4, 512)), cap["gen.rb4.out"])
    e_z = rel_linf(gen.last_latent(), cap["z"])
    _record(S=S, B=B, precision=precision + ("+fp6 main-conv cross pieces (MSR_F16C_FP6=1)" if os.environ.get("MSR_F16C_FP6") == "1" else ""),
            oracle=oracle_dtype, rel_linf_output=err, rel_linf_rb5_x1=e_x1,
            rb5_mean1=e_mean, rb5_std1=e_std, rel_linf_rb4_out=e_rb4, rel_linf_z=e_z)
    gen.close()
    del gen
    torch.cuda.empty_cache()
    assert y.shape == (B, S, S, 1) and np.isfinite(y).all()
    assert err <= TOL, err
    assert max(e_x1, e_rb4, e_z, e_mean, e_std) <= TOL, (e_x1, e_rb4, e_z, e_mean, e_std)


def test_f16c_with_the_opt_in_fp6_cross_pieces_matches_oracle():
    """MSR_F16C_FP6=1 (read once per process): the stream-kernel main convs take fp6 e2m3 cross pieces with block scales
    (kernels.h PREC_F16C6), written by the gamma|beta convs' epilogues.  Same bar; the BASELINE shapes run in a child process."""
    import subprocess
    import sys
    env = dict(os.environ, MSR_F16C_FP6="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_baseline_config_matches_oracle and f16c and (256-16 or 512-8)"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("switch", ["MSR_GBR=0", "MSR_TILE_WALK=0", "MSR_PP_KSPLIT=0", "MSR_F16C_KSPLIT=0", "MSR_F16C_SW=0",
                                    "MSR_F16C_SW=2", "MSR_FUSE_MOMENTS=0", "MSR_SMALLCIN_TILED=0"])
def test_f16c_parity_under_every_documented_kernel_switch(switch):
    """The behaviour-changing environment switches of the library (each read once per process: A/B dispatch of the conv kernels,
    tile walk, K ranges) must all leave the default mode inside the parity bar: GauGAN(256, 16) against the oracle in a child
    process per switch.  (MSR_F16C_FP6=1 has its own test above.)"""
    import subprocess
    import sys
    k, v = switch.split("=")
    env = dict(os.environ, **{k: v})
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_baseline_config_matches_oracle and f16c and 256-16"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("precision", ["f16c", "bf16x3", "fp32"])
@pytest.mark.parametrize("variant", ["gaugan", "gaugan_no_kl"])
def test_spade64_golden_in_both_precisions(hip_lib, variant, precision):
    """The committed S = 64 golden vectors (fp64 oracle) and per-block checksums, for each arithmetic explicitly."""
    from moonsuperresolution_amd import Generator
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"spade64_{variant}.npz"))
    w = make_weights(variant, 64, seed=1234, bias_scale=0.05)
    gen = Generator(64, 2, variant=variant, weights=w, eps=make_latent_noise(2, 256, 7), precision=precision)
    y = gen(synthetic_patches(2, 64, 0), training=False)
    assert rel_linf(y, g["output"]) <= TOL
    for i in range(1, 7):
        r = 1 << (i - 1)
        out = gen.debug_tensor(f"ws.gen.rb{i}.out", (2, r, r, GEN_FILTERS[i - 1]))
        assert abs(np.abs(out).mean() / g[f"gen_rb{i}_out"] - 1) <= TOL, i
    gen.close()


def test_clone_and_pipeline_follow_loaded_weights(hip_lib):
    """Generator.clone() carries the weights the handle holds NOW (after load()), and the tiler's second pipeline
    handle follows a load() made after the tiler was built: pipeline=2 equals pipeline=1 bit for bit."""
    from moonsuperresolution_amd import DEMSuperResolution, DSRConfig, Generator
    from tests.helpers import synthetic_raster
    w_a = make_weights("gaugan_no_kl", 64, seed=1, bias_scale=0.05)
    w_b = make_weights("gaugan_no_kl", 64, seed=2, bias_scale=0.05)
    x = torch.from_numpy(synthetic_patches(4, 64, 0)).cuda()
    gen = Generator(64, 4, variant="gaugan_no_kl", weights=w_a)
    y_a = gen.forward_device(x).cpu().numpy()
    cfg = DSRConfig(image_size=64, stride=32, batch_size=4, tile_size=128)
    d2 = DEMSuperResolution(cfg, model=gen, pipeline=2)      # clones the seed-1 weights here
    gen.load(w_b)
    y_b = gen.forward_device(x).cpu().numpy()
    assert not np.array_equal(y_a, y_b)
    twin = gen.clone()
    assert np.array_equal(twin.forward_device(x).cpu().numpy(), y_b)
    twin.close()
    img, dem = synthetic_raster(120, 120, 9)
    m2, s2, g2 = d2.processMap(img, dem)
    d1 = DEMSuperResolution(cfg, model=gen, pipeline=1)
    m1, s1, g1 = d1.processMap(img, dem)
    assert np.array_equal(m1, m2) and np.array_equal(s1, s2) and np.array_equal(g1, g2)
    d1.close(); d2.close(); gen.close()


@pytest.mark.parametrize("precision", ["f16c", "bf16x3"])
def test_last_batch_of_a_tile_one_patch_and_seven_zero_patches(hip_lib, precision):
    """process_full_tiles.py:468-474 at the production geometry: 529 patches in batches of 8 leave ONE real patch, padded
    with seven all-zero patches that take part in SPADE's batch statistics (spade.py:21) — the most lopsided batch the
    path produces (zero patches make the mask embedding, and with it gamma and beta, constant over 7/8 of the batch).
    GauGAN(512, 8) against the fp32 oracle on exactly that batch."""
    from moonsuperresolution_amd import Generator
    from oracle import generator_ref
    S, B = 512, 8
    key = ("zeros", S, B)
    if key not in _ORACLE:
        _ORACLE.clear()
        w = make_weights("gaugan", S, seed=1234, bias_scale=0.05)
        eps = make_latent_noise(B, 256, 7)
        x = synthetic_patches(B, S, 0)
        x[1:] = 0.0
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        _ORACLE[key] = (x, w, eps, np.asarray(generator_ref.spade_call(x, w, "gaugan", eps, dtype=torch.float32), np.float64))
    x, w, eps, ref = _ORACLE[key]
    gen = Generator(S, B, variant="gaugan", weights=w, eps=eps, precision=precision)
    y = gen(x, training=False)
    err = rel_linf(y, ref)
    err_real = rel_linf(y[0], ref[0])
    _record(S=S, B=B, precision=precision, oracle="float32", batch="1 real + 7 zero patches", rel_linf_output=err,
            rel_linf_real_patch=err_real)
    gen.close()
    del gen
    torch.cuda.empty_cache()
    assert np.isfinite(y).all() and err <= TOL and err_real <= TOL, (err, err_real)


F16_TOL = 3e-3     # declared (VERDICT r2 item 3 asked for <= 3e-2 for a usable configs[4] mode).  Measured 8.9e-4 / 8.0e-4
                   # relative L-inf, 6.3e-4 rms at (256,16) / (512,8): inside north_star's 1e-3 at these shapes, but with a 1.1x
                   # margin only — which is why "f16" is a declared-tolerance mode and "f16c" (4e-5) stays the default


@pytest.mark.parametrize("S,B", [(256, 16), (512, 8)])
def test_f16_mode_declared_tolerance(hip_lib, S, B):
    """precision="f16" (MSR_FLAG_F16_MAIN): the f16c data path with the cross terms left out of conv_gb_resident and
    conv_igemm_f16c_sw — ONE fp16 product per element.  A single fp8 product cannot meet 3e-2 on this network whatever the
    scaling recipe (profiles/r03_fp8_mode_emulation.txt: 0.09-0.26), a single fp16 product can: the relative L-infinity
    against the oracle is measured, recorded and bounded by F16_TOL.  Not a parity mode (north_star: 1e-3)."""
    from moonsuperresolution_amd import Generator
    x, w, eps, ref, cap, oracle_dtype = _oracle(S, B)
    gen = Generator(S, B, variant="gaugan", weights=w, eps=eps, precision="f16")
    y = gen(x, training=False)
    err = rel_linf(y, ref)
    rms = float(np.sqrt(np.mean((np.asarray(y, np.float64) - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))
    _record(S=S, B=B, precision="f16", oracle=oracle_dtype, rel_linf_output=err, rel_rms_output=rms,
            note="declared-tolerance mode (one fp16 product per element)")
    gen.close()
    del gen
    torch.cuda.empty_cache()
    assert np.isfinite(y).all() and err <= F16_TOL, err


FP8_TOL = 0.25     # declared, NON-parity (measured 0.16 relative L-inf, 0.13 relative rms at both BASELINE sizes)


@pytest.mark.parametrize("S,B", [(256, 16), (512, 8)])
def test_fp8_mode_declared_tolerance(hip_lib, S, B):
    """BASELINE configs[4] ("fp16 with fp8 MFMA conv"): precision="fp8" runs the chip-filling gamma|beta and ResidualBlock
    convs on fp8 e4m3 weights x bf8 e5m2 activations (3 and 2 mantissa bits).  It is NOT a parity mode: its relative
    L-infinity against the oracle is measured here, printed, appended to gpurun_out/parity_baseline_configs.jsonl and
    bounded by FP8_TOL — the tolerance this mode declares instead of north_star's 1e-3.  The kernels themselves are
    exact on quantised operands (tests/test_gpu_conv_kernel.py::test_conv_fp8_exact_on_quantised_operands)."""
    from moonsuperresolution_amd import Generator
    x, w, eps, ref, cap, oracle_dtype = _oracle(S, B)
    gen = Generator(S, B, variant="gaugan", weights=w, eps=eps, precision="fp8")
    y = gen(x, training=False)
    err = rel_linf(y, ref)
    rms = float(np.sqrt(np.mean((np.asarray(y, np.float64) - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))
    sw = S // 64
    e_rb4 = rel_linf(gen.debug_tensor("ws.gen.rb4.out", (B, sw * 8, sw * 8, 512)), cap["gen.rb4.out"])
    _record(S=S, B=B, precision="fp8", oracle=oracle_dtype, rel_linf_output=err, rel_rms_output=rms, rel_linf_rb4_out=e_rb4,
            note="declared non-parity mode")
    gen.close()
    del gen
    torch.cuda.empty_cache()
    assert np.isfinite(y).all() and err <= FP8_TOL, err
