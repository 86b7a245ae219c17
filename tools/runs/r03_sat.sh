#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_conv_kernel.py -x -q -m gpu -k "saturation or resident or head" -s 2>&1 | tail -15
bash tools/runs/r03_bench_quick.sh
