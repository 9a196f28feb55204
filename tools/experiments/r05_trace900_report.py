"""Per kernel of a rocprofv3 --kernel-trace CSV: mean duration over the first and the last quarter of its launches, and every
flush duration in launch order:  python tools/experiments/r05_trace900_report.py p_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").replace("satrans::", "").split("(")[0][:40]
    by[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, v in by.items():
    if "flush" in name:
        print(name, "durations us:", " ".join(f"{x:.0f}" for x in v))
    elif len(v) >= 10:
        q = len(v) // 4
        print(f"{name:42s} n={len(v):5d} mean first quarter {sum(v[:q]) / q:8.1f}  last quarter {sum(v[-q:]) / q:8.1f}")
