"""Per-family busy time of a rocprofv3 --kernel-trace CSV over a time window: total kernel time, union of the
intervals (what the GPU was busy with at least one kernel of the family), and the idle time no kernel covers.
usage: python profiles/trace_families.py <kernel_trace.csv> [skip_fraction_at_start | tiles]
"tiles": the window runs from the second patch_stats launch (the first belongs to the warm-up tile) to the end of the last
stitch_tile launch, i.e. the timed tile loop of raster_bench.py."""
import csv
import sys

path = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "0.0"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
if mode == "tiles":
    ps = [r[0] for r in rows if "patch_stats" in r[2]]
    st = [r[1] for r in rows if "stitch_tile" in r[2]]
    w0, t1 = ps[1], st[-1]
else:
    w0 = t0 + int((t1 - t0) * float(mode))
rows = [r for r in rows if r[0] >= w0 and r[1] <= t1]
span = (t1 - w0) / 1e6


def fam(name):
    for key in ("conv_igemm_f16c_sw", "conv_igemm_bf16x3_pp", "conv_igemm_bf16x3_halo", "conv_igemm_bf16x3", "conv_igemm", "splitk_epilogue",
                "conv_smallcin", "moments", "norm_act", "dense", "latent", "head_kernel", "patch_stats", "extract_patches",
                "compact_patches", "stitch", "fillBuffer", "copyBuffer"):
        if key in name:
            return key
    return name.split("(")[0][-40:]


def union(iv):
    iv.sort()
    tot, ca, cb = 0, None, None
    for a, b in iv:
        if cb is None or a > cb:
            if cb is not None:
                tot += cb - ca
            ca, cb = a, b
        else:
            cb = max(cb, b)
    return tot + (cb - ca if cb is not None else 0)


by = {}
for a, b, n in rows:
    by.setdefault(fam(n), []).append((a, b))
print(f"window {span:.2f} ms, {len(rows)} kernels")
print(f"{'family':28s} {'launches':>9s} {'sum ms':>10s} {'union ms':>10s} {'% of window':>12s}")
for k, iv in sorted(by.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
    s = sum(b - a for a, b in iv) / 1e6
    u = union(list(iv)) / 1e6
    print(f"{k:28s} {len(iv):9d} {s:10.3f} {u:10.3f} {100 * u / span:11.2f}%")
busy = union([(a, b) for a, b, _ in rows]) / 1e6
print(f"any kernel running: {busy:.2f} ms = {100 * busy / span:.2f}% of the window; idle {span - busy:.2f} ms")
gaps = []
iv = sorted((a, b) for a, b, _ in rows)
cb = iv[0][1]
for a, b in iv[1:]:
    if a > cb:
        gaps.append((a - cb) / 1e3)
    cb = max(cb, b)
gaps.sort(reverse=True)
print("largest idle gaps (us):", [round(g, 1) for g in gaps[:12]], " count >50us:", sum(g > 50 for g in gaps))
