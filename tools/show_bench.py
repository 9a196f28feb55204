"""Print the headline and per-phase numbers of a bench.py JSON line read from stdin."""
import json
import sys

d = json.load(sys.stdin)
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["unit"], "|", d["ms_per_step"], "ms/step")
for k, v in d["kernels"].items():
    print("   ", k, v["ms_per_step"], "ms/step", v.get("achieved", ""), v.get("unit", ""), v.get("frac", ""))
if d.get("cpu_baseline"):
    print("    cpu_baseline", d["cpu_baseline"])
