#!/bin/bash
# the "f16" declared-tolerance mode: error test at both BASELINE sizes, bench line, and the default mode's bench beside it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_f16; mkdir -p $O
python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu -k "f16_mode or (matches_oracle and f16c and 256)" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
grep '"f16"' gpurun_out/parity_baseline_configs.jsonl | tail -2
for prec in f16 f16c; do
python bench.py --no-cpu-baseline --no-also --steps 40 --warmup 10 --precision $prec 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $prec', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k: round(v,3) for k,v in d['kernel_ms_per_call'].items()})"
done
