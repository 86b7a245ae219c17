import csv, sys
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        d[r["Name"][:90]]=(int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
    return d
a=load(sys.argv[1]); b=load(sys.argv[2])
print(f"{'kernel':92s} {'calls':>6s} {'fast avg us':>12s} {'slow avg us':>12s} {'ratio':>6s} {'d total ms':>10s}")
for k in sorted(a, key=lambda k:-a[k][2])[:22]:
    if k in b:
        print(f"{k:92s} {a[k][0]:6d} {a[k][1]/1e3:12.1f} {b[k][1]/1e3:12.1f} {b[k][1]/a[k][1]:6.3f} {(b[k][2]-a[k][2])/1e6:10.1f}")
