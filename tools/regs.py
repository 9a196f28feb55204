"""Diagnostic: registers, spills and LDS of every kernel in a -save-temps assembly file (…gfx950.s).
    python tools/regs.py <file.s> [name-substring]"""
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else ""
meta = s[s.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda f: (re.search(r"\." + f + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if key not in name:
        continue
    agpr = blk.split("\n", 1)[0].strip()
    m = re.search(r"I(.*?)EEv", name)
    short = name.split("I", 1)[0].replace("_ZN7satrans", "")[2:] + "<" + (m.group(1) if m else "") + ">"
    short = short.replace("ELi", ",").replace("ELb", ",b").replace("Li", "").replace("Lb", "b")
    print(f"{short:70s} vgpr {g('vgpr_count'):>4s} agpr {agpr:>4s} spill {g('vgpr_spill_count'):>4s} sgpr_spill {g('sgpr_spill_count'):>3s} "
          f"scratch {g('private_segment_fixed_size'):>5s}")
