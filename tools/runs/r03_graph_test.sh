cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_generator.py -m gpu -x -q -k graph 2>&1 | tail -15
