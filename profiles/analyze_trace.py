"""Per-layer TFLOP/s of the conv_igemm_f32 launches from a rocprofv3 --kernel-trace CSV.
usage: python profiles/analyze_trace.py <kernel_trace.csv> <S> <B>"""
import csv, sys
sys.path.insert(0, ".")
from moonsuperresolution_amd.weights import ENC_CHANNELS, GEN_FILTERS

path, S, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
plan = []
cin, r = 64, S // 2
for i, c in enumerate(ENC_CHANNELS[1:], start=2):
    r //= 2
    plan.append((f"enc.ds{i} s2", r, cin, c)); cin = c
sw = S // 64; cin = 1024
for i, f in enumerate(GEN_FILTERS, start=1):
    r = sw << (i - 1)
    learned = f != cin
    plan.append((f"rb{i}.gb1", r, 128, 2 * cin)); plan.append((f"rb{i}.conv1", r, cin, f))
    if learned:
        plan.append((f"rb{i}.gb3", r, 128, 2 * cin)); plan.append((f"rb{i}.conv3", r, cin, f))
    plan.append((f"rb{i}.gb2", r, 128, 2 * f)); plan.append((f"rb{i}.conv2", r, f, f))
    cin = f
rows = [r for r in csv.DictReader(open(path)) if "conv_igemm" in r["Kernel_Name"] or "conv_gb_resident" in r["Kernel_Name"]]   # split-K epilogues excluded
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(plan)
calls = len(rows) // n
last = rows[(calls - 1) * n: calls * n]
tot_ms = tot_fl = 0
print(f"{'layer':14s} {'r':>4s} {'Cin':>5s} {'N':>5s} {'grid':>7s} {'wg':>4s} {'ms':>8s} {'TF/s':>7s}  kernel")
for (name, r, ci, N), row in zip(plan, last):
    ms = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    fl = 2.0 * B * r * r * ci * N * 9
    tot_ms += ms; tot_fl += fl
    k = row["Kernel_Name"].split("(")[0].replace("void msr::", "")
    print(f"{name:14s} {r:4d} {ci:5d} {N:5d} {int(row['Grid_Size_X'])//int(row['Workgroup_Size_X']):7d} {row['Workgroup_Size_X']:>4s} {ms:8.3f} {fl/ms/1e9:7.1f}  {k}")
print(f"total conv {tot_ms:.2f} ms, {tot_fl/1e9:.1f} GFLOP, {tot_fl/tot_ms/1e9:.1f} TF/s")
