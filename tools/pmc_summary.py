"""Summarise the rocprofv3 --pmc passes of tools/pmc_passes.sh: per kernel, mean counter value per launch.

    python tools/pmc_summary.py gpurun_out/pmc aliccp > profiles/rNN_pmc_summary.json

Keys: the kernel name without its argument list.  The fused backward kernel is launched in two instantiations per step since
round 4 - `layer_bwd_fused_kernel` (layers L-2 .. 0) and `layer_bwd_fused_kernel[head]` (the last layer with the head fused in,
template parameter HEADF) - and both fused layer kernels are also listed per layer (dispatch order), because layer 0 reads its rows
through the fused gather."""
import collections
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
LAYERS = int(os.environ.get("SATRANS_PMC_LAYERS", "3"))   # bench --config aliccp: forward l0 .. l(L-2), backward [head] l(L-1), then l(L-2) .. l0
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))


def canonical(name):
    short = name.split("(")[0].replace("void ", "").replace("satrans::", "").strip()
    m = re.match(r"(layer_(?:fwd|bwd)_fused_kernel)<(.*)>$", short)
    if not m:
        return short[:64], None
    args = [a.strip() for a in m.group(2).split(",")]
    if m.group(1) == "layer_bwd_fused_kernel" and len(args) >= 9 and args[8] == "true":
        return "layer_bwd_fused_kernel[head]", short
    return m.group(1), short


inst = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        rows = sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"]))
    seen = collections.defaultdict(lambda: collections.defaultdict(int))      # per kernel and counter: dispatches so far
    for row in rows:
        k = row.get("Kernel_Name") or row.get("Kernel Name")
        c, v = row["Counter_Name"], float(row["Counter_Value"])
        short, full = canonical(k)
        if full:
            inst.setdefault(short, set()).add(full)
        a = acc[short][c]
        a[0] += v
        a[1] += 1
        layer = None
        if short == "layer_fwd_fused_kernel":
            per = max(1, LAYERS - 1)      # (the last layer has no forward launch: its forward is recomputed inside [head])
            layer = seen[short][c] % per
        elif short == "layer_bwd_fused_kernel":
            per = max(1, LAYERS - 1)
            layer = per - 1 - (seen[short][c] % per)
        elif short == "layer_bwd_fused_kernel[head]":
            layer = LAYERS - 1
        if layer is not None:
            seen[short][c] += 1
            b = acc[f"{short} @layer{layer}"][c]
            b[0] += v
            b[1] += 1
            b_t = acc[f"{short} @layer{layer}"]["duration_ns"]
            b_t[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            b_t[1] += 1
out = {}
for k, cs in acc.items():
    out[k] = {c: a[0] / a[1] for c, a in cs.items()}
    out[k]["launches"] = max(a[1] for a in cs.values())
    if k in inst:
        out[k]["instantiations"] = sorted(inst[k])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satrans_amd import native  # noqa: E402
keep = [k for k in out if "layer_" in k or "gather_rows" in k or "lazy_" in k or "touched" in k or "head_kernel" in k
        or "sort_fields" in k or "bucket_" in k or "reduce" in k or k.startswith("gen_")]
res = {k: out[k] for k in sorted(keep)}
# provenance: the hash of the kernel SOURCES (two builds of the same sources give different .so bytes) and the bench config
res["_source_sha256"] = native.source_hash()
res["_config"] = sys.argv[2] if len(sys.argv) > 2 else "aliccp"
res["_products"] = "f32"
print(json.dumps(res, indent=1))
