"""Diagnostic: instruction census of a kernel per barrier-delimited segment (compile with -save-temps first).
    python tools/asm_census.py <file.s> <mangled-name-substring>"""
import collections
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
start = s.index(key + ":") if (key + ":") in s else s.index(key)
start = s.index("\n", start)
end = s.index(".end_amdhsa_kernel", start)
seg = 0
counts = collections.defaultdict(collections.Counter)
for l in s[start:end].split("\n"):
    l = l.strip()
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        continue
    op = l.split()[0]
    if op == "s_barrier":
        seg += 1
    counts[seg][op] += 1
for k in sorted(counts):
    c = counts[k]
    tot = sum(c.values())
    if tot < 25:
        continue
    pick = lambda pre: sum(v for o, v in c.items() if o.startswith(pre))
    print(f"seg {k:2d}: total {tot:5d} valu {pick('v_') - pick('v_mfma'):5d} mfma {pick('v_mfma'):4d} salu {pick('s_'):4d} "
          f"ds {pick('ds_'):4d} scratch ld/st {pick('scratch_load'):3d}/{pick('scratch_store'):3d} "
          f"global {pick('global_'):3d} nop {c['s_nop']:4d} lane {c['v_readlane_b32'] + c['v_writelane_b32']:4d}")
