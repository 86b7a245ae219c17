"""Durations of the launches whose name contains a pattern, in launch order (rocprofv3 --kernel-trace CSV).
usage: python profiles/list_kernels.py <kernel_trace.csv> <pattern> [last N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if sys.argv[2] in r["Kernel_Name"]]
for r in sel[-(int(sys.argv[3]) if len(sys.argv) > 3 else 12):]:
    print(r["Kernel_Name"][:48], "grid", r.get("Grid_Size_X"), r.get("Grid_Size_Y"), "wg", r.get("Workgroup_Size_X"),
          f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.1f} us')
