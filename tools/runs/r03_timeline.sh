#!/bin/bash
# timelines of one call (B=1 and B=8 at S=512), B=1 latency with graphs, and VERDICT item 5's exact halo command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_timeline; mkdir -p $O
rocprofv3 --kernel-trace -d $O/b1 -o t --output-format csv -- python3 profiles/run_forwards_b1.py 512 6 f16c > $O/b1.log 2>&1 || { tail -5 $O/b1.log; exit 1; }
python profiles/call_timeline.py $O/b1/t_kernel_trace.csv 6 > $O/timeline_b1.txt
rocprofv3 --kernel-trace -d $O/b8 -o t --output-format csv -- python3 profiles/run_forwards.py spade512 4 f16c > $O/b8.log 2>&1 || { tail -5 $O/b8.log; exit 1; }
python profiles/call_timeline.py $O/b8/t_kernel_trace.csv 4 > $O/timeline_b8.txt
python profiles/analyze_trace.py $O/b8/t_kernel_trace.csv 512 8 > $O/conv_layers_b8.txt
tail -3 $O/timeline_b1.txt $O/timeline_b8.txt
python - <<'PY'
import statistics, time, torch
from moonsuperresolution_amd import Generator, make_latent_noise, make_weights, synthetic_patches
S = 512
gen = Generator(S, 1, variant="gaugan", weights=make_weights("gaugan", S, seed=1234), eps=make_latent_noise(1, 256, 7), precision="f16c")
x = torch.from_numpy(synthetic_patches(1, S, seed=0)).cuda(); out = torch.empty((1, S, S, 1), device="cuda")
with torch.cuda.stream(torch.cuda.Stream()):
    for _ in range(6): gen.forward_device(x, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(40):
        t = time.perf_counter(); gen.forward_device(x, out=out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print("B=1 S=512 f16c p50 ms:", round(statistics.median(ts), 3), "min", round(min(ts), 3))
PY
timeout -k 10 500 python raster_bench.py --halo --rows 15000 --cols 70000 --image-size 512 --stride 64 --batch-size 8 --simulate-rank 3 --simulate-world 8 --max-rows 12 > $O/halo_item5.json 2> $O/halo_item5.err || { tail -20 $O/halo_item5.err; exit 1; }
cat $O/halo_item5.json
