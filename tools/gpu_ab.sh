#!/bin/bash
# A/B of two builds of libmoonsr_hip.so on ONE box (boxes differ by ~5 %): alternates bench.py runs.
# usage: tools/gpu_ab.sh <libA.so> <libB.so> [rounds] [extra bench args]
A=$1; B=$2; R=${3:-2}; shift 3
for i in $(seq 1 $R); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    MSR_LIB=$lib timeout -k 10 250 python bench.py --steps 60 --warmup 10 "$@" > gpurun_out/ab_$v$i.log 2>&1 || exit 1
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v$i.log").read().strip().splitlines()[-1])
print("$v$i", round(d["value"],1), round(d["ms_per_step"],3), round(d.get("kernel_ms_per_call",{}).get("conv_igemm_bf16x3",0),3), flush=True)
PY
  done
done
