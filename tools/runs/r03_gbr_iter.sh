set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
LVL=3 CFGS="256:256" bash tools/runs/r03_gbr_stamps.sh || exit 1
