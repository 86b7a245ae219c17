#!/bin/bash
# incremental build of libmoonsr_hip.so + refresh of the content stamp (so that the GPU box does not rebuild it)
cd /root/repo || exit 1
make -C moonsuperresolution_amd/csrc -j8 2>&1 | grep -E "error|Error" -A3
test -f moonsuperresolution_amd/csrc/libmoonsr_hip.so || exit 1
python - <<'PY'
from moonsuperresolution_amd import _lib
open(_lib.STAMP_PATH, "w").write(_lib.sources_hash() + "\n")
print("stale:", _lib.is_stale())
PY
