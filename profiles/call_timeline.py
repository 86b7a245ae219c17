"""Every kernel of the LAST generator call of a rocprofv3 --kernel-trace CSV in start order: start offset, duration, and
the idle gap before it (one handle; the aux stream's kernels overlap the main stream's, which shows as a negative gap).
usage: python profiles/call_timeline.py <kernel_trace.csv> <calls in the run>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "msr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
calls = int(sys.argv[2])
n = len(rows) // calls
last = rows[(calls - 1) * n:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
busy = 0
print(f"{len(last)} launches in the last of {calls} calls")
print(f"{'start us':>9s} {'dur us':>8s} {'gap us':>8s} {'grid':>7s} {'wg':>4s}  kernel")
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].split("(")[0].replace("void msr::", "").replace("msr::", "")
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:8.1f} "
          f"{int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) // int(r['Workgroup_Size_X']):7d} {r['Workgroup_Size_X']:>4s}  {k}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"call span {(prev_end - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
