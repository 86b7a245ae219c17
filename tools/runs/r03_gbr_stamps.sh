# usage: [LVL=1|2] [CFGS="r:C r:C ..."] bash tools/runs/r03_gbr_stamps.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf /tmp/st && mkdir -p /tmp/st/pkg && cp -r moonsuperresolution_amd/csrc /tmp/st/pkg/csrc && cp -r include /tmp/st/include || exit 1
make -C /tmp/st/pkg/csrc clean > /dev/null 2>&1
make -C /tmp/st/pkg/csrc -j16 all EXTRA="-DMSR_DIAG_BUILD -DMSR_GB_STAMPS=${LVL:-1}" > gpurun_out/r03_gbr_stamps_build.log 2>&1 || { tail -5 gpurun_out/r03_gbr_stamps_build.log; exit 1; }
for cfg in ${CFGS:-256:256 256:128 128:512 64:1024}; do
  r=${cfg%%:*}; C=${cfg##*:}
  MSR_ALLOW_DIAG_BUILD=1 MSR_LIB=/tmp/st/pkg/csrc/libmoonsr_hip.so timeout -k 10 200 python tools/gpu_gbr_stamps.py $r $C > gpurun_out/r03_gbr_stamps_${r}_$C.txt 2>&1 || { tail -5 gpurun_out/r03_gbr_stamps_${r}_$C.txt; exit 1; }
  echo "== r=$r C=$C (N=$((2*C)))"; grep "gbr wave 0" gpurun_out/r03_gbr_stamps_${r}_$C.txt | tail -23
done
