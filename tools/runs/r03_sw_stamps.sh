# in-kernel s_memtime stamps of the stream kernel, per phase (E, O, C) of the tap pairs: long-K main conv and the Cin = 128 shape
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf /tmp/st && mkdir -p /tmp/st/pkg && cp -r moonsuperresolution_amd/csrc /tmp/st/pkg/csrc && cp -r include /tmp/st/include || exit 1
for lvl in 1 2; do
  make -C /tmp/st/pkg/csrc clean > /dev/null 2>&1
  make -C /tmp/st/pkg/csrc -j16 all EXTRA="-DMSR_DIAG_BUILD -DMSR_SW_STAMPS=$lvl -DSW_UNROLL2=0" > gpurun_out/r03_stamps_build_$lvl.log 2>&1 || { tail -5 gpurun_out/r03_stamps_build_$lvl.log; exit 1; }
  for kind in main gb; do
    MSR_ALLOW_DIAG_BUILD=1 MSR_LIB=/tmp/st/pkg/csrc/libmoonsr_hip.so timeout -k 10 200 python tools/gpu_sw_stamps.py $kind > gpurun_out/r03_sw_stamps_${kind}_$lvl.txt 2>&1 || exit 1
    echo "== level $lvl $kind"; tail -40 gpurun_out/r03_sw_stamps_${kind}_$lvl.txt | grep -E "wave 0" | tail -32
  done
done
