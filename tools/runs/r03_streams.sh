#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for st in 1 2 3 2 1; do python bench.py --no-cpu-baseline --no-also --steps 48 --warmup 12 --streams $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $st', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"; done
