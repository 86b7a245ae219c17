"""Per-LAYER table of rocprofv3 --pmc counters for the conv launches of the LAST generator call of a run of
profiles/run_forwards.py (one handle, one stream: the conv launches of a call come in plan order).

usage: python profiles/pmc_by_layer.py <S> <B> <counter_collection.csv> [more counter_collection.csv ...]

Every CSV is one PMC pass of the same command; counters of all passes are joined per layer.  FETCH_SIZE / WRITE_SIZE
are converted to bytes as MI355X_MICROARCH.md "HBM" prescribes (KiB counters; FETCH_SIZE doubled on gfx950) and put
next to the layer's algorithmic bytes (input tensor once + x / residual once + output once + weights once; fp32
words, which is also the size of the f16c chunk images).  Infinity-Cache hits are counted by these counters.
"""
import collections
import csv
import sys

sys.path.insert(0, ".")
from moonsuperresolution_amd.weights import ENC_CHANNELS, GEN_FILTERS  # noqa: E402


def layer_plan(S, B):
    plan = []
    cin, r = 64, S // 2
    for i, c in enumerate(ENC_CHANNELS[1:], start=2):
        r //= 2
        # strided encoder conv: input at 2r
        plan.append(dict(name=f"enc.ds{i} s2", r=r, cin=cin, n=c, in_px=B * (2 * r) ** 2, aux=0))
        cin = c
    sw = S // 64
    cin = 1024
    for i, f in enumerate(GEN_FILTERS, start=1):
        r = sw << (i - 1)
        learned = f != cin
        px = B * r * r
        # gamma|beta conv: reads the 128-channel embedding and x (C = N/2 channels, at r/2 when up-sampled), writes C
        def gb(tag, c):
            return dict(name=f"rb{i}.{tag}", r=r, cin=128, n=2 * c, in_px=px, aux=px * c // (4 if tag != "gb2" and i > 1 else 1),
                        out_c=c)
        plan.append(gb("gb1", cin))
        plan.append(dict(name=f"rb{i}.conv1", r=r, cin=cin, n=f, in_px=px, aux=0))
        if learned:
            plan.append(gb("gb3", cin))
            plan.append(dict(name=f"rb{i}.conv3", r=r, cin=cin, n=f, in_px=px, aux=0))
        plan.append(gb("gb2", f))
        # conv2 adds the residual (block input, up-sampled index) or the skip branch
        plan.append(dict(name=f"rb{i}.conv2", r=r, cin=f, n=f, in_px=px, aux=px * f))
        cin = f
    for L in plan:
        out_c = L.get("out_c", L["n"])
        L["flop"] = 2.0 * B * L["r"] ** 2 * L["cin"] * L["n"] * 9
        L["alg_bytes"] = 4.0 * (L["in_px"] * L["cin"] + L["aux"] + B * L["r"] ** 2 * out_c + 9 * L["cin"] * L["n"])
    return plan


def read_pass(path):
    """{dispatch id: (kernel, {counter: value}, start, end)} for the conv_igemm launches."""
    d = {}
    for r in csv.DictReader(open(path)):
        if "conv_igemm" not in r["Kernel_Name"] and "conv_gb_resident" not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        e = d.setdefault(k, [r["Kernel_Name"].split("(")[0].replace("void msr::", ""), collections.defaultdict(float),
                             int(r["Start_Timestamp"]), int(r["End_Timestamp"])])
        e[1][r["Counter_Name"]] += float(r["Counter_Value"])
    return d


if __name__ == "__main__":
    S, B = int(sys.argv[1]), int(sys.argv[2])
    plan = layer_plan(S, B)
    n = len(plan)
    per_layer = [dict(kernel="", ms=[]) for _ in plan]
    counters = []
    for path in sys.argv[3:]:
        d = read_pass(path)
        ids = sorted(d)
        calls = len(ids) // n
        if calls < 1 or len(ids) % n:
            print(f"# {path}: {len(ids)} conv launches is not a multiple of {n} — skipped", file=sys.stderr)
            continue
        for L, k in zip(per_layer, ids[(calls - 1) * n:]):
            kern, vals, t0, t1 = d[k]
            L["kernel"] = kern
            L["ms"].append((t1 - t0) / 1e6)
            for c, v in vals.items():
                L[c] = v
                if c not in counters:
                    counters.append(c)
    hdr = f"{'layer':12s} {'r':>4s} {'Cin':>5s} {'N':>5s} {'ms':>7s} {'alg MB':>8s}"
    has_traffic = "FETCH_SIZE" in counters and "WRITE_SIZE" in counters
    if has_traffic:
        hdr += f" {'read MB':>8s} {'write MB':>8s} {'x alg':>6s} {'TB/s':>5s}"
    other = [c for c in counters if c not in ("FETCH_SIZE", "WRITE_SIZE")]
    for c in other:
        hdr += f" {c[-22:]:>22s}"
    print(hdr + "  kernel")
    tot_alg = tot_rd = tot_wr = 0.0
    for P, L in zip(plan, per_layer):
        ms = sum(L["ms"]) / max(len(L["ms"]), 1)
        line = f"{P['name']:12s} {P['r']:4d} {P['cin']:5d} {P['n']:5d} {ms:7.3f} {P['alg_bytes'] / 1e6:8.1f}"
        if has_traffic:
            rd, wr = 2.0 * L.get("FETCH_SIZE", 0.0) * 1024, L.get("WRITE_SIZE", 0.0) * 1024
            tot_alg += P["alg_bytes"]; tot_rd += rd; tot_wr += wr
            line += f" {rd / 1e6:8.1f} {wr / 1e6:8.1f} {(rd + wr) / P['alg_bytes']:6.2f} {(rd + wr) / ms / 1e9 if ms else 0:5.2f}"
        for c in other:
            line += f" {L.get(c, float('nan')):22.0f}"
        print(line + "  " + L["kernel"])
    if has_traffic:
        print(f"conv family per call: algorithmic {tot_alg / 1e9:.2f} GB, read {tot_rd / 1e9:.2f} GB, written {tot_wr / 1e9:.2f} GB, "
              f"ratio {(tot_rd + tot_wr) / tot_alg:.2f} (Infinity-Cache hits are counted as traffic by these counters)")
