set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_conv_kernel.py -m gpu -x -q -k resident 2>&1 | tail -3 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -k "test_baseline_config_matches_oracle and f16c" 2>&1 | tail -3 || exit 1
LVL=2 CFGS="256:256" bash tools/runs/r03_gbr_stamps.sh || exit 1
