cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_ab.sh build/libmoonsr_prev.so moonsuperresolution_amd/csrc/libmoonsr_hip.so 3 --no-cpu-baseline --no-also
