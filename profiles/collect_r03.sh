#!/bin/bash
# Round-3 measurement artefacts (run through gpurun from the repo root).  Outputs land in gpurun_out/collect_r03/;
# the summaries that are judged are copied into profiles/ with an r03_ prefix.  PART=a|b|c splits the work over calls.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/collect_r03; mkdir -p $O
PART=${1:-abc}
if [[ $PART == *a* ]]; then
# per-layer tables (known call count, one stream, launches in plan order)
for cfg in "spade512 f16c" "spade512 f16" "spade512 fp8" "spade512 bf16x3" "spade256 f16c"; do
  set -- $cfg; wl=$1; pr=$2; S=${wl#spade}; B=16; [ $S = 512 ] && B=8
  rocprofv3 --kernel-trace --stats -d $O/kt_${wl}_$pr -o kt --output-format csv -- python3 profiles/run_forwards.py $wl 12 $pr > $O/kt_${wl}_$pr.log 2>&1 || exit 1
  python profiles/analyze_trace.py $O/kt_${wl}_$pr/kt_kernel_trace.csv $S $B > $O/${wl}_${pr}_conv_layers.txt || echo "layer table failed for $cfg"
  cp $O/kt_${wl}_$pr/kt_kernel_stats.csv $O/${wl}_${pr}_kernel_stats.csv
  echo "trace $wl $pr done"
done
python profiles/call_timeline.py $O/kt_spade512_f16c/kt_kernel_trace.csv 12 > $O/timeline_b8.txt
rocprofv3 --kernel-trace --stats -d $O/kt_b1 -o kt --output-format csv -- python3 profiles/run_forwards_b1.py 512 12 f16c > $O/kt_b1.log 2>&1 || exit 1
python profiles/call_timeline.py $O/kt_b1/kt_kernel_trace.csv 12 > $O/timeline_b1.txt
cp $O/kt_b1/kt_kernel_stats.csv $O/spade512_b1_f16c_kernel_stats.csv
echo "B=1 trace done"
fi
if [[ $PART == *b* ]]; then
# HBM traffic: separate PMC passes per mode
for pr in f16c f16 fp8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $O/pmc_${pr}_$c -o p --output-format csv -- python3 profiles/run_forwards.py spade512 3 $pr > $O/pmc_${pr}_$c.log 2>&1 || exit 1
  done
  python profiles/summarize_pmc.py $O/pmc_${pr}_FETCH_SIZE/p_counter_collection.csv $O/pmc_${pr}_WRITE_SIZE/p_counter_collection.csv spade512_$pr $O/spade512_${pr}_pmc_summary.json 3 > /dev/null || exit 1
  echo "pmc traffic $pr done"
done
python profiles/pmc_by_layer.py 512 8 $O/pmc_f16c_FETCH_SIZE/p_counter_collection.csv $O/pmc_f16c_WRITE_SIZE/p_counter_collection.csv > $O/spade512_f16c_traffic_by_layer.txt || echo "traffic table failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq_f16c -o p --output-format csv -- python3 profiles/run_forwards.py spade512 3 f16c > $O/pmc_sq_f16c.log 2>&1 || exit 1
python profiles/pmc_util.py $O/pmc_sq_f16c/p_counter_collection.csv conv > $O/spade512_f16c_sq_counters.txt || echo "sq table failed"
echo "pmc sq done"
fi
if [[ $PART == *c* ]]; then
# the driver's own command under the kernel trace: its per-kernel averages must agree with bench.py's event timing
rocprofv3 --kernel-trace --stats -d $O/kt_bench -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-also > $O/kt_bench.log 2>&1 || exit 1
cp $O/kt_bench/kt_kernel_stats.csv $O/bench_py_kernel_stats.csv
echo "trace of bench.py done"
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
python bench.py --streams 2 --no-cpu-baseline --no-also > $O/bench_2streams.json 2> /dev/null || exit 1
echo "bench done"
python raster_bench.py > $O/raster256.json 2> /dev/null || exit 1
python raster_bench.py --image-size 512 --stride 64 --batch-size 8 > $O/raster512.json 2> /dev/null || exit 1
(python tools/gpu_determinism_soak.py 512 8 100 && python tools/gpu_determinism_soak.py 256 16 150) > $O/determinism_soak.txt 2>&1 || exit 1
echo "all done"
fi
