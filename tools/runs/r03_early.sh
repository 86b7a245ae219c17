#!/bin/bash
# early-phase work (fused moments, smallcin prefetch, dense double buffer): parity tests, timelines, B=1 latency, bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_early; mkdir -p $O
python -m pytest tests/test_gpu_generator.py tests/test_gpu_conv_kernel.py -x -q -m gpu -k "${TESTK:-not resident}" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
rocprofv3 --kernel-trace -d $O/b1 -o t --output-format csv -- python3 profiles/run_forwards_b1.py 512 6 f16c > $O/b1.log 2>&1 || { tail -5 $O/b1.log; exit 1; }
python profiles/call_timeline.py $O/b1/t_kernel_trace.csv 6 > $O/timeline_b1.txt
rocprofv3 --kernel-trace -d $O/b8 -o t --output-format csv -- python3 profiles/run_forwards.py spade512 4 f16c > $O/b8.log 2>&1 || { tail -5 $O/b8.log; exit 1; }
python profiles/call_timeline.py $O/b8/t_kernel_trace.csv 4 > $O/timeline_b8.txt
python profiles/analyze_trace.py $O/b8/t_kernel_trace.csv 512 8 > $O/conv_layers_b8.txt
tail -n 2 $O/timeline_b1.txt; tail -n 2 $O/timeline_b8.txt
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
bash tools/runs/r03_bench_quick.sh
