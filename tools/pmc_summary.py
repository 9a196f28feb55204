"""Summarise the rocprofv3 --pmc passes of tools/pmc_passes.sh: per kernel, mean counter value per launch."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
LAYERS = 3          # bench --config aliccp: the layer kernels run in the order l0 l1 l2 (forward), l2 l1 l0 (backward) every step
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        rows = sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"]))
    seen = collections.defaultdict(lambda: collections.defaultdict(int))      # per kernel and counter: dispatches so far
    for row in rows:
        k = row.get("Kernel_Name") or row.get("Kernel Name")
        c, v = row["Counter_Name"], float(row["Counter_Value"])
        short = k.split("(")[0].replace("void ", "").replace("satrans::", "")[:48]
        a = acc[short][c]
        a[0] += v
        a[1] += 1
        # the fused layer kernels also per layer: layer 0 gathers its rows from the embedding arena (fused gather)
        if short.startswith(("layer_fwd_fused_kernel", "layer_bwd_fused_kernel")):
            i = seen[short][c] % LAYERS
            seen[short][c] += 1
            layer = i if short.startswith("layer_fwd") else LAYERS - 1 - i
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
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satrans_amd import native  # noqa: E402
keep = [k for k in out if "layer_" in k or "gather_rows" in k or "lazy_" in k or "touched" in k or "head_kernel" in k
        or k.startswith("gen_")]
res = {k: out[k] for k in sorted(keep)}
# provenance: the hash of the kernel SOURCES (two builds of the same sources give different .so bytes) and the bench config
res["_source_sha256"] = native.source_hash()
res["_config"] = sys.argv[2] if len(sys.argv) > 2 else "aliccp"
res["_products"] = "split" if native.lib().satrans_get_product_mode() == 1 else "f32"   # (SATRANS_PRODUCTS of the profiled run = of this process)
print(json.dumps(res, indent=1))
