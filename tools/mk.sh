#!/bin/bash
# incremental build of libmoonsr_hip.so + refresh of the content stamp (so that the GPU box does not rebuild it)
cd /root/repo || exit 1
if ! make -C moonsuperresolution_amd/csrc -j8 > /tmp/mk.log 2>&1; then grep -E "error" -A3 /tmp/mk.log | head -40; echo "BUILD FAILED"; exit 1; fi
python - <<'PY'
from moonsuperresolution_amd import _lib
open(_lib.STAMP_PATH, "w").write(_lib.sources_hash() + "\n")
print("stale:", _lib.is_stale())
PY
