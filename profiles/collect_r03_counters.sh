#!/bin/bash
# Round-3 evidence for the conv kernels' co-limiters: vector-memory (TA / TCP),
# LDS and L2 counters per LAYER, plus HBM traffic per layer.  One PMC pass per counter group (gfx950: 8 SQ slots,
# 4 TCC, 2 GRBM per pass; TA / TCP kept to a few per pass), each on `run_forwards.py spade512 3 <mode>`.
# usage (through gpurun, from the repo root): bash profiles/collect_r03_counters.sh [mode] [tag]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
MODE=${1:-f16c}; TAG=${2:-$MODE}
O=gpurun_out/collect_r03_$TAG; mkdir -p $O
pass() {   # name, counters...
  local name=$1; shift
  if rocprofv3 --pmc "$@" --kernel-trace -d $O/pmc_$name -o p --output-format csv -- python3 profiles/run_forwards.py spade512 3 $MODE > $O/pmc_$name.log 2>&1; then
    echo "pass $name ok"
  else
    echo "pass $name FAILED (see $O/pmc_$name.log)"; tail -5 $O/pmc_$name.log
  fi
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sqmem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass sqmfma SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass ta1 TA_TA_BUSY_sum TA_BUSY_max GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass ta3 TA_BUFFER_LOAD_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tcp2 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcp3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
ls $O/pmc_*/p_counter_collection.csv > $O/passes.txt 2>/dev/null
python profiles/pmc_by_layer.py 512 8 $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv > $O/traffic_by_layer.txt || echo "traffic table failed"
python profiles/pmc_by_layer.py 512 8 $O/pmc_sqmem/p_counter_collection.csv > $O/sqmem_by_layer.txt || echo "sqmem table failed"
python profiles/pmc_by_layer.py 512 8 $O/pmc_sqmfma/p_counter_collection.csv > $O/sqmfma_by_layer.txt || echo "sqmfma table failed"
python profiles/pmc_by_layer.py 512 8 $O/pmc_ta1/p_counter_collection.csv $O/pmc_ta2/p_counter_collection.csv $O/pmc_ta3/p_counter_collection.csv > $O/ta_by_layer.txt || echo "ta table failed"
python profiles/pmc_by_layer.py 512 8 $O/pmc_tcp1/p_counter_collection.csv $O/pmc_tcp2/p_counter_collection.csv $O/pmc_tcp3/p_counter_collection.csv $O/pmc_tcc/p_counter_collection.csv > $O/tcp_tcc_by_layer.txt || echo "tcp table failed"
echo "counters done"
