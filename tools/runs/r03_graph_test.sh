cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 1100 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_conv_kernel.py tests/test_gpu_generator.py -m gpu -x -q 2>&1 | tail -12
