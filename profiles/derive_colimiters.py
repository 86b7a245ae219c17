"""Derived per-layer utilisation of the conv kernels from the PMC passes of profiles/collect_r03_counters.sh.

usage: python profiles/derive_colimiters.py <S> <B> <dir with pmc_*/p_counter_collection.csv>

Formulas (MI355X_MICROARCH.md: SQ_* cycle counters are summed over the chip, GRBM_GUI_ACTIVE over the 8 XCDs, TA / TCP
`_sum` counters over the 256 CUs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAIT_* quad-cycles per wave):
  clk GHz   = GRBM_GUI_ACTIVE / 8 / kernel time          (reads high on dispatches shorter than ~0.3 ms)
  mfma %    = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
  ta %      = TA_TA_BUSY_sum / 256 / (GRBM_GUI_ACTIVE / 8)            (texture addresser = the vector-memory issue path)
  ta_stl %  = TA_ADDR_STALLED_BY_TC_CYCLES_sum / 256 / (GRBM / 8)     (address path held by the L1)
  lds %     = SQ_LDS_IDX_ACTIVE / 256 / (GRBM / 8);  lds_cf % = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  l2hit %   = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  l1->l2 lat= TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum     (cycles per L1 miss)
  vmem/k    = SQ_INSTS_VMEM_RD per 1000 MFMA-busy cycles of one SIMD
"""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_by_layer import layer_plan, read_pass  # noqa: E402

S, B, root = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
plan = layer_plan(S, B)
n = len(plan)
rows = [dict(ms=[]) for _ in plan]
for path in sorted(glob.glob(os.path.join(root, "pmc_*", "p_counter_collection.csv"))):
    d = read_pass(path)
    ids = sorted(d)
    if not ids or len(ids) % n:
        continue
    for L, k in zip(rows, ids[(len(ids) // n - 1) * n:]):
        kern, vals, t0, t1 = d[k]
        L["kernel"] = kern
        L["ms"].append((t1 - t0) / 1e6)
        if "GRBM_GUI_ACTIVE" in vals:       # ratios are taken inside ONE pass: counters and its own GRBM
            for c, v in vals.items():
                L[c + "/grbm"] = v / vals["GRBM_GUI_ACTIVE"]
        for c, v in vals.items():
            L[c] = v


def g(L, k, default=float("nan")):
    return L.get(k, default)


print(f"{'layer':12s} {'ms':>6s} {'TF/s':>6s} {'clk':>5s} {'mfma%':>6s} {'ta%':>5s} {'ta_stl%':>7s} {'lds%':>5s} {'lds_cf%':>7s} {'l2hit%':>6s} "
      f"{'l1>l2 lat':>9s} {'xalg':>5s}  kernel")
for P, L in zip(plan, rows):
    ms = sum(L["ms"]) / max(len(L["ms"]), 1)
    grbm8 = g(L, "GRBM_GUI_ACTIVE") / 8
    # TA / TCP / TCC passes carried no GRBM counter of their own: use this layer's cycles from the passes that did
    # (same kernel, same box, same clock within a few per cent)
    mfma = g(L, "SQ_VALU_MFMA_BUSY_CYCLES/grbm") * 8 / 1024
    ta = g(L, "TA_TA_BUSY_sum/grbm") * 8 / 256 if "TA_TA_BUSY_sum/grbm" in L else g(L, "TA_TA_BUSY_sum") / 256 / grbm8
    ta_st = g(L, "TA_ADDR_STALLED_BY_TC_CYCLES_sum") / 256 / grbm8
    lds = g(L, "SQ_LDS_IDX_ACTIVE/grbm") * 8 / 256
    cf = g(L, "SQ_LDS_BANK_CONFLICT") / max(g(L, "SQ_LDS_IDX_ACTIVE", 1.0), 1.0)
    hit = g(L, "TCC_HIT_sum") / max(g(L, "TCC_HIT_sum", 0.0) + g(L, "TCC_MISS_sum", 0.0), 1.0)
    lat = g(L, "TCP_TCC_READ_REQ_LATENCY_sum") / max(g(L, "TCP_TCC_READ_REQ_sum", 1.0), 1.0)
    traffic = (2.0 * g(L, "FETCH_SIZE", 0.0) + g(L, "WRITE_SIZE", 0.0)) * 1024
    print(f"{P['name']:12s} {ms:6.3f} {P['flop'] / ms / 1e9:6.0f} {grbm8 / ms / 1e6:5.2f} {100 * mfma:6.1f} {100 * ta:5.1f} {100 * ta_st:7.1f} "
          f"{100 * lds:5.1f} {100 * cf:7.2f} {100 * hit:6.1f} {lat:9.0f} {traffic / P['alg_bytes']:5.2f}  {L.get('kernel', '')}")
