cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest_a.log 2>&1; rc=$?
tail -15 gpurun_out/r03_gputest_a.log; exit $rc
