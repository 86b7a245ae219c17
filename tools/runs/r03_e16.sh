#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_conv_kernel.py -x -q -m gpu -k "resident or saturation" -s 2>&1 | grep -v amdgpu.ids | tail -8
python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "matches_oracle and f16c" 2>&1 | tail -3
grep '"f16c"' gpurun_out/parity_baseline_configs.jsonl | tail -3
bash tools/gpu_ab.sh build/libmoonsr_prev.so moonsuperresolution_amd/csrc/libmoonsr_hip.so 3 --no-cpu-baseline --no-also
