"""Diagnostic: min / median / max duration of every kernel whose name contains a substring, from a rocprofv3 --kernel-trace CSV.
    python tools/kernel_minmax.py <p_kernel_trace.csv> [substring, default layer_bwd]"""
import csv
import statistics
import sys
from collections import defaultdict

key = sys.argv[2] if len(sys.argv) > 2 else "layer_bwd"
d = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if key in r["Kernel_Name"]:
        d[r["Kernel_Name"].split("(")[0][-60:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(f"{k:60s} n={len(v):4d} min {min(v):7.1f} median {statistics.median(v):7.1f} max {max(v):7.1f} us")
