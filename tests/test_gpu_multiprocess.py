"""GPU (-m gpu): the N > 1 path of the REAL tile loop in separate processes (process_full_tiles.py:313-325,431-479 sharded by
tile row; the halo mode of moonsuperresolution_amd/halo.py), rehearsed on one device: two fresh child ranks of
torch.distributed.run over gloo, both on device 0 (raster_bench.py's MSR_BENCH_BACKEND / MSR_BENCH_DEVICE knobs; RCCL needs
one GPU per rank, which the test box does not have).  What must hold bit for bit:
  * tile-row sharding: the rows gathered by the two ranks == the rows of one process (batch composition per tile is the
    reference's in both, so SPADE's batch statistics are identical);
  * halo mode: the two ranks' gathered canvas == the two-rank run simulated inside one process (same accumulate / exchange /
    finish code, the exchange done by torch.distributed send / recv instead of a hand-over of tensors).
The children are started as new processes (never an exec of this process, which has touched the GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_raster(tmp_path, name, gpus, extra):
    out = str(tmp_path / f"{name}.npz")
    env = dict(os.environ, MSR_BENCH_BACKEND="gloo", MSR_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "raster_bench.py"), "--gpus", str(gpus), "--no-reference", "--dump", out] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
    return np.load(out), json.loads(line)


def test_two_process_tile_rows_equal_one_process(tmp_path):
    args = ["--rows", "2048", "--cols", "3072", "--image-size", "256", "--stride", "32", "--batch-size", "16"]
    one, info1 = run_raster(tmp_path, "one", 1, args)
    two, info2 = run_raster(tmp_path, "two", 2, args + ["--gather"])
    assert info1["n_gpus"] == 1 and info2["n_gpus"] == 2 and info2["gathered"]
    assert info1["tiles_total"] == info2["tiles_total"] == 6 and info1["patches"] == info2["patches"] > 0
    for k in ("mean", "std", "good"):
        assert one[k].shape == two[k].shape and np.array_equal(one[k], two[k], equal_nan=True), k
    assert one["good"].any()


def test_two_process_halo_mode_equals_the_simulated_two_rank_run(tmp_path, hip_lib):
    import torch
    from moonsuperresolution_amd import DSRConfig, Generator, HaloShardedSuperResolution
    sys.path.insert(0, ROOT)
    from raster_bench import synthetic_raster
    rows, cols, S, s, B, T = 1100, 900, 128, 32, 8, 256
    got, info = run_raster(tmp_path, "halo2", 2, ["--rows", str(rows), "--cols", str(cols), "--image-size", str(S), "--stride", str(s),
                                                   "--batch-size", str(B), "--tile-size", str(T), "--halo", "--gather"])
    assert info["n_gpus"] == 2 and info["patches"] > 0
    img, dem = synthetic_raster(rows, cols, seed=0)
    gen = Generator(S, B, variant="gaugan", weights=1234, eps=7)
    d = HaloShardedSuperResolution(DSRConfig(image_size=S, stride=s, batch_size=B, tile_size=T), model=gen)
    d.setImages(img, dem)
    states = [d.haloAccumulate(r, 2) for r in range(2)]
    slabs = [d.haloFinish(states[0], None, states[1]["send_down"]), d.haloFinish(states[1], states[0]["send_up"], None)]
    want = [torch.cat([sl[0][k] for sl in slabs], dim=0).cpu().numpy() for k in range(3)]
    for k, name in enumerate(("mean", "std", "good")):
        assert got[name].shape == want[k].shape and np.array_equal(got[name], want[k], equal_nan=True), name
    assert got["good"].any()
    d.close(); gen.close()
