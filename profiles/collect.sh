#!/bin/bash
# Collect the round's measurement artefacts on a GPU box (run through gpurun from the repo root):
#   kernel-trace + stats, per-layer conv tables, HBM-traffic PMC passes (separate runs, as the guide prescribes),
#   SQ utilisation counters, bench JSON lines for both workloads and precisions, raster and latency probes.
# Outputs land in gpurun_out/collect/; copy the summaries you want judged into profiles/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/collect; mkdir -p $O
for wl in spade256 spade512; do
  S=${wl#spade}; B=16; [ $S = 512 ] && B=8
  rocprofv3 --kernel-trace --stats -d $O/kt_$wl -o kt --output-format csv -- python3 profiles/run_forwards.py $wl 25 > $O/kt_$wl.log 2>&1 || exit 1
  python profiles/analyze_trace.py $O/kt_$wl/kt_kernel_trace.csv $S $B > $O/${wl}_bf16x3_conv_layers.txt || exit 1
  echo "trace $wl done"
done
rocprofv3 --kernel-trace --stats -d $O/kt_fp32 -o kt --output-format csv -- python3 profiles/run_forwards.py spade256 13 fp32 > $O/kt_fp32.log 2>&1 || exit 1
python profiles/analyze_trace.py $O/kt_fp32/kt_kernel_trace.csv 256 16 > $O/spade256_fp32_conv_layers.txt || exit 1
echo "trace fp32 done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o p --output-format csv -- python3 profiles/run_forwards.py spade256 3 > $O/pmc_$c.log 2>&1 || exit 1
done
python profiles/summarize_pmc.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv spade256_bf16x3 $O/spade256_bf16x3_pmc_summary.json 3 > /dev/null || exit 1
echo "pmc traffic done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -o p --output-format csv -- python3 profiles/run_forwards.py spade256 3 > $O/pmc_sq.log 2>&1 || exit 1
python profiles/pmc_util.py $O/pmc_sq/p_counter_collection.csv conv > $O/spade256_bf16x3_sq_counters.txt || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_lds -o p --output-format csv -- python3 profiles/run_forwards.py spade256 3 > $O/pmc_lds.log 2>&1 || exit 1
python profiles/pmc_util.py $O/pmc_lds/p_counter_collection.csv conv >> $O/spade256_bf16x3_sq_counters.txt || exit 1
echo "pmc sq done"
rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt --output-format csv -- python3 bench.py --no-cpu-baseline > $O/kt_bench.log 2>&1 || exit 1
echo "trace of bench.py done"
python bench.py > $O/bench_spade256_bf16x3.json 2> $O/bench_spade256_bf16x3.err || exit 1
python bench.py --streams 2 --no-cpu-baseline > $O/bench_spade256_bf16x3_2streams.json 2> /dev/null || exit 1
python bench.py --workload spade512 > $O/bench_spade512_bf16x3.json 2> $O/bench_spade512_bf16x3.err || exit 1
python bench.py --precision fp32 --steps 20 > $O/bench_spade256_fp32.json 2> $O/bench_spade256_fp32.err || exit 1
echo "bench done"
python tools/gpu_latency.py > $O/latency.txt 2>&1 || exit 1
python tools/gpu_p2p_latency.py > $O/p2p_latency.txt 2>&1 || exit 1
python raster_bench.py > $O/raster.txt 2>&1 || exit 1
python raster_bench.py --image-size 512 --stride 64 --batch-size 8 > $O/raster512.txt 2>&1 || exit 1
echo "all done"
