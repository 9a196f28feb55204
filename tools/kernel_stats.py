"""Print a rocprofv3 --kernel-trace --stats summary (…_kernel_stats.csv) as a table: name, calls, average, share."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print(f"{'kernel':72s} {'calls':>6s} {'avg us':>10s} {'share':>8s}")
for r in rows[:top]:
    name = r["Name"].replace("void satrans::", "").replace("satrans::", "")[:72]
    print(f"{name:72s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.1f} {float(r['TotalDurationNs']) / tot * 100:7.2f}%")
