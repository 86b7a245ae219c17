cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do python bench.py --no-cpu-baseline --no-also --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), {k: round(v,3) for k,v in d['kernel_ms_per_call'].items()})"; done
