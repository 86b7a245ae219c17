set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_conv_kernel.py tests/test_gpu_baseline_configs.py -m gpu -x -q > gpurun_out/r03_walk_tests.log 2>&1 || { tail -20 gpurun_out/r03_walk_tests.log; exit 1; }
tail -3 gpurun_out/r03_walk_tests.log
for i in 1 2; do
  MSR_TILE_WALK=0 python bench.py --no-cpu-baseline --no-also --steps 40 --warmup 10 > gpurun_out/r03_walk_off_$i.json 2>/dev/null || exit 1
  python bench.py --no-cpu-baseline --no-also --steps 40 --warmup 10 > gpurun_out/r03_walk_on_$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json
for n in ("off_1","on_1","off_2","on_2"):
    d=json.loads(open(f"gpurun_out/r03_walk_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"],1), round(d["ms_per_step"],3), d["kernel_ms_per_call"])
PY
O=gpurun_out/collect_r03_walk; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p --output-format csv -- python3 profiles/run_forwards.py spade512 3 f16c > $O/pmc_$c.log 2>&1 || exit 1
done
python profiles/pmc_by_layer.py 512 8 $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv > $O/traffic_by_layer.txt
tail -1 $O/traffic_by_layer.txt
