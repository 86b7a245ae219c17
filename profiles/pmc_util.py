"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name (sum of counter values over dispatches).
usage: python profiles/pmc_util.py <counter_collection.csv> [name-filter]"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msr::", "")
    if flt and flt not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items()):
    n = max(cnt[(k, c)] for c in d)
    print(f"{k}  dispatches={n}")
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {v:18.0f}  per-dispatch {v / n:14.0f}")
