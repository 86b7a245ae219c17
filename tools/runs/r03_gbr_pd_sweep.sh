# prefetch-depth sweep of conv_gb_resident: in-kernel stamps (level 2) per variant
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf /tmp/st && mkdir -p /tmp/st/pkg && cp -r moonsuperresolution_amd/csrc /tmp/st/pkg/csrc && cp -r include /tmp/st/include || exit 1
make -C /tmp/st/pkg/csrc clean > /dev/null 2>&1
make -C /tmp/st/pkg/csrc -j16 all EXTRA="-DMSR_DIAG_BUILD -DMSR_GB_STAMPS=2" > gpurun_out/r03_pd_build.log 2>&1 || exit 1
for v in "6 3" "8 3" "6 5" "8 5" "4 2" "10 6"; do
  set -- $v
  rm -f /tmp/st/pkg/csrc/conv_gbr.o /tmp/st/pkg/csrc/libmoonsr_hip.so
  make -C /tmp/st/pkg/csrc -j16 all EXTRA="-DMSR_DIAG_BUILD -DMSR_GB_STAMPS=2 -DGB_PD=$1 -DGB_PDC=$2" > gpurun_out/r03_pd_build.log 2>&1 || { tail -5 gpurun_out/r03_pd_build.log; exit 1; }
  MSR_ALLOW_DIAG_BUILD=1 MSR_LIB=/tmp/st/pkg/csrc/libmoonsr_hip.so timeout -k 10 200 python tools/gpu_gbr_stamps.py 256 256 > gpurun_out/r03_pd_$1_$2.txt 2>&1 || { tail -5 gpurun_out/r03_pd_$1_$2.txt; exit 1; }
  echo "PD=$1 PDC=$2:"; for w in 0 3; do grep "gbr wave $w" gpurun_out/r03_pd_$1_$2.txt | tail -23 | sed -n 13,23p | awk '{printf "%s ", $6}'; echo; done
done
