"""Summarise tools/pmc_c5.sh: HBM bytes of the general layer path (BASELINE configs[4]) per layer forward / backward CHAIN and per
kernel family.  A layer of that path is ~25 (forward) / ~45 (backward) launches; dispatches are attributed by their position in
the step: gen_* launches between the scenario-table forward and the head are forward, those between the head and the
touched-row / scenario-table backward kernels are backward.

    python tools/pmc_c5_summary.py gpurun_out/pmc_c5 LAYERS > profiles/rNN_c5_pmc_summary.json

Bytes = 2 x FETCH_SIZE + WRITE_SIZE in KiB x 1024 (the gfx950 correction of MI355X_MICROARCH.md), per launch of the chain."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_c5"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def short(name):
    return name.split("(")[0].replace("void ", "").replace("satrans::", "").strip()[:72]


per_counter = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        rows = sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        per_counter.setdefault(r["Counter_Name"], []).append((short(r.get("Kernel_Name") or r.get("Kernel Name")), float(r["Counter_Value"])))

out = {}
for counter, seq in per_counter.items():
    state, steps = "pre", 0
    chain = collections.defaultdict(float)
    fam = collections.defaultdict(lambda: [0.0, 0])
    for name, v in seq:
        if name.startswith("scenario_table_fwd"):
            state = "fwd"
            steps += 1
        elif name.startswith("head_kernel"):
            state = "head"
        elif name.startswith("head_reduce"):
            state = "bwd"
        elif name.startswith(("scenario_table_bwd", "touched_", "adam_", "lazy_flush")):
            state = "tail"
        if name.startswith("gen_") and state in ("fwd", "bwd"):
            chain[state] += v
        f = fam[name]
        f[0] += v
        f[1] += 1
    out[counter] = {"steps": steps, "layer_fwd_chain": chain["fwd"] / max(1, steps * L), "layer_bwd_chain": chain["bwd"] / max(1, steps * L),
                    "per_kernel_mean": {k: a[0] / a[1] for k, a in sorted(fam.items()) if k.startswith(("gen_", "lazy_", "touched", "head"))},
                    "per_kernel_launches": {k: a[1] for k, a in sorted(fam.items()) if k.startswith(("gen_", "lazy_", "touched", "head"))}}
res = {"layers": L}
f, w = out.get("FETCH_SIZE"), out.get("WRITE_SIZE")
for key in ("layer_fwd_chain", "layer_bwd_chain"):
    if f and w:
        res[key] = {"FETCH_SIZE": f[key], "WRITE_SIZE": w[key], "bytes_per_launch": round((2.0 * f[key] + w[key]) * 1024.0)}
if f and w:
    res["per_kernel_bytes_per_launch"] = {k: round((2.0 * f["per_kernel_mean"][k] + w["per_kernel_mean"].get(k, 0.0)) * 1024.0)
                                          for k in f["per_kernel_mean"]}
    res["per_kernel_launches_per_step"] = {k: v / max(1, f["steps"]) for k, v in f["per_kernel_launches"].items()}
    res["steps_profiled"] = f["steps"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satrans_amd import native  # noqa: E402
res["_source_sha256"] = native.source_hash()
res["_config"] = "c5"
print(json.dumps(res, indent=1))
