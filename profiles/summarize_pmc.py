"""Summarise rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs) into HBM bytes per
launch per kernel family.  Units and corrections follow MI355X_MICROARCH.md "HBM": both counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of a wide coalesced read stream, so it is doubled; WRITE_SIZE is exact.

usage: python profiles/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <workload> <out.json> [forwards]
(forwards = generator calls in each profiled run, warm-up included: bench.py --steps 2 --warmup 1 -> 3)
"""
import collections
import csv
import json
import sys


def family(name: str) -> str:
    name = name.split("(")[0]
    if "conv_igemm" in name or "splitk_epilogue" in name or "conv_gb_resident" in name:
        return "conv_igemm"      # every instantiation of the implicit-GEMM conv (+ its split-K epilogue) and, since round 3, the
                                 # resident SPADE-layer kernel (the library's profiler counts it in the same family)
    return name.replace("void ", "").replace("msr::", "").split("<")[0]


def collect(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        f = family(r["Kernel_Name"])
        agg[f][0] += 1
        agg[f][1] += float(r["Counter_Value"])
    return agg


if __name__ == "__main__":
    fetch, write, workload, out = sys.argv[1:5]
    forwards = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    fa, wa = collect(fetch, "FETCH_SIZE"), collect(write, "WRITE_SIZE")
    res = {"workload": workload, "forwards": forwards,
           "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (KiB counters, gfx950 FETCH correction); the conv_igemm family "
                   "includes the split-K epilogue kernels; bench.py divides hbm_bytes_per_forward by its conv ops per call"}
    for fam in sorted(set(fa) | set(wa)):
        if not fam.startswith(("conv", "moments", "head", "dense", "norm", "latent")):
            continue
        n = max(fa[fam][0], wa[fam][0])
        rd = 2.0 * fa[fam][1] * 1024 / max(fa[fam][0], 1)
        wr = wa[fam][1] * 1024 / max(wa[fam][0], 1)
        res[fam] = {"kernel_launches": n, "hbm_read_bytes_per_kernel_launch": rd,
                    "hbm_write_bytes_per_kernel_launch": wr, "hbm_bytes_per_kernel_launch": rd + wr,
                    "hbm_bytes_per_forward": (2.0 * fa[fam][1] + wa[fam][1]) * 1024 / forwards}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
