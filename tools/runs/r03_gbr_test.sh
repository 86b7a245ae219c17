set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_conv_kernel.py -m gpu -x -q -k "resident" -s > gpurun_out/r03_gbr_test.log 2>&1; rc=$?
tail -30 gpurun_out/r03_gbr_test.log; exit $rc
