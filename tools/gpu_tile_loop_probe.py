"""Probe (not a pytest): which ingredient of the tile loop costs throughput with two generator streams.
usage: python tools/gpu_tile_loop_probe.py [S B stride]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ".")
import torch

from moonsuperresolution_amd import DEMSuperResolution, DSRConfig, Generator, _lib
from raster_bench import synthetic_raster

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
s = int(sys.argv[3]) if len(sys.argv) > 3 else S // 8
img, dem = synthetic_raster(2048, 2048)
gen = Generator(S, B, weights=1234, eps=7)
d = DEMSuperResolution(DSRConfig(image_size=S, stride=s, batch_size=B, tile_size=1024), model=gen, pipeline=2)
d.setImages(img, dem)
d.padInputs()
tiles = d.generateTileList()
d.processTile(*tiles[0])
torch.cuda.synchronize()
ncall = d.last_counts[1]
st = d._prepare_tile(*tiles[0])
st["event"].synchronize()
sx, sy, mm = st["sx"], st["sy"], st["mm_sel"]
rows, cols = d.dem_padded_shape
preds = d._bufs[0]["preds"]
lib, h = d._lib, d._h


def run(name, per_call_out, extract, nstreams=2, reps=2, gated=False):
    outs = [torch.empty((B, S, S, 1), device="cuda") for _ in range(2)]
    best = 0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cur = torch.cuda.current_stream()
        for ps in d._pstreams:
            ps.wait_stream(cur)
        evs = [torch.cuda.Event() for _ in range(4)]
        last = None
        for c in range(ncall):
            k = c % nstreams
            with torch.cuda.stream(d._pstreams[k]):
                if extract:
                    lib.msr_extract_patches(h, d.img_padded.data_ptr(), d.dem_padded.data_ptr(), rows, cols,
                                            sx[c * B:].data_ptr(), sy[c * B:].data_ptr(), mm[c * B:].data_ptr(), B,
                                            d._batches[k].data_ptr(), torch.cuda.current_stream().cuda_stream)
                out = preds[c * B:(c + 1) * B].unsqueeze(-1) if per_call_out else outs[k]
                d._gens[k].forward_device(d._batches[k], out=out, gate=last if gated else None)
                if gated:
                    last = evs[c % 4]
                    last.record(d._pstreams[k])
        torch.cuda.synchronize()
        best = max(best, ncall * B * (S / 512.0) ** 2 / (time.perf_counter() - t0))
    print(f"{name:50s} {best:8.1f} tiles512/s", flush=True)


def full(tag):
    t0 = time.perf_counter()
    n = 0
    for key, out in d.iterTiles(tiles):
        n += d.last_counts[1]
    torch.cuda.synchronize()
    print(f"{'iterTiles (full loop, 4 tiles) ' + tag:50s} {n * B * (S / 512.0) ** 2 / (time.perf_counter() - t0):8.1f} tiles512/s", flush=True)


if os.environ.get("PROBE_WARM"):
    gen.forward_device(torch.zeros((B, S, S, 2), device="cuda"))
    torch.cuda.synchronize()
full("first")
full("second")
run("2 streams, fixed in/out", False, False)
run("2 streams, fixed in/out, GATED", False, False, gated=True)
run("2 streams, extract + preds slice, GATED", True, True, gated=True)
run("2 streams, out = preds slice", True, False)
run("2 streams, extract + fixed out", False, True)
run("2 streams, extract + preds slice", True, True)
run("1 stream, extract + preds slice", True, True, nstreams=1)
full("last")
run("2 streams, fixed in/out (again)", False, False)
