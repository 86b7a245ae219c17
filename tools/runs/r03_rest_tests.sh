cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 1100 python -m pytest tests/test_gpu_halo.py tests/test_gpu_multiprocess.py tests/test_gpu_preprocess.py tests/test_gpu_tiler.py -m gpu -x -q 2>&1 | tail -15
