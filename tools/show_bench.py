"""Print the headline and per-phase numbers of a bench.py JSON line: `show_bench.py FILE [label]` (or `-` for stdin)."""
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "-"
d = json.load(sys.stdin if path == "-" else open(path))
print(sys.argv[2] if len(sys.argv) > 2 else path, d["value"], d["unit"], "|", d["ms_per_step"], "ms/step")
for k, v in d["kernels"].items():
    print("   ", k, v["ms_per_step"], "ms/step", v.get("achieved", ""), v.get("unit", ""), v.get("frac", ""))
if d.get("cpu_baseline"):
    print("    cpu_baseline", d["cpu_baseline"])
