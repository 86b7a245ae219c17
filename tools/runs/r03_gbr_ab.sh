set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
  MSR_GBR=0 python bench.py --no-cpu-baseline --no-also --steps 40 --warmup 10 > gpurun_out/r03_gbr_off_$i.json 2>/dev/null || exit 1
  python bench.py --no-cpu-baseline --no-also --steps 40 --warmup 10 > gpurun_out/r03_gbr_on_$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json
for n in ("off_1","on_1","off_2","on_2"):
    d=json.loads(open(f"gpurun_out/r03_gbr_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"],1), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), {k: round(v,3) for k,v in d["kernel_ms_per_call"].items()}, d.get("p50_ms_per_call_b1"), d.get("p50_ms_per_call_b1_eager"))
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/kt_gbr -o kt --output-format csv -- python3 profiles/run_forwards.py spade512 10 f16c > gpurun_out/kt_gbr.log 2>&1 || exit 1
python profiles/list_kernels.py gpurun_out/kt_gbr/kt_kernel_stats.csv 2>/dev/null | head -20 || head -20 gpurun_out/kt_gbr/kt_kernel_stats.csv
