#!/bin/bash
# halo mode after the band rewrite: GPU tests, then rank 3 of 8 of the 15000 x 70000 raster for 24 patch rows
set -e
cd /root/repo
python -m pytest tests/test_gpu_halo.py tests/test_gpu_multiprocess.py -x -q -m gpu 2>&1 | tail -5
mkdir -p gpurun_out
timeout -k 10 600 python raster_bench.py --halo --rows 15000 --cols 70000 --simulate-rank 3 --simulate-world 8 --max-rows 24 --band-rows 2 > gpurun_out/r03_halo_rank3of8.json 2> gpurun_out/r03_halo_rank3of8.err || { tail -20 gpurun_out/r03_halo_rank3of8.err; exit 1; }
cat gpurun_out/r03_halo_rank3of8.json
