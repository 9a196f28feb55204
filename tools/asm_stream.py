"""Diagnostic: the instruction stream of a kernel's barrier-delimited segments as one character per instruction
(M mfma, r ds_read, w ds_write, g global/buffer, . other VALU, , SALU, n s_nop, [..] s_waitcnt) - shows at a glance whether
LDS reads are batched ahead of the matrix instructions or funnelled one wait per product.
    python tools/asm_stream.py <file.s> <mangled-name-substring> seg [seg ...]"""
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
start = s.index(key)
start = s.index("\n", s.index(":", start))
end = s.index(".end_amdhsa_kernel", start)
seg, segs = 0, {}
for l in s[start:end].split("\n"):
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_barrier":
        seg += 1
    segs.setdefault(seg, []).append(t)
for k in [int(v) for v in sys.argv[3:]]:
    out = []
    for t in segs.get(k, []):
        op = t.split()[0]
        if op.startswith("v_mfma"):
            out.append("M")
        elif op.startswith("ds_read") or op.startswith("ds_load"):
            out.append("r")
        elif op.startswith("ds_write") or op.startswith("ds_store"):
            out.append("w")
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"):
            out.append("g")
        elif op == "s_waitcnt":
            out.append("[" + t.split(None, 1)[1].replace("lgkmcnt", "L").replace("vmcnt", "V").replace(" ", "") + "]")
        elif op == "s_nop":
            out.append("n")
        elif op.startswith("v_"):
            out.append(".")
        else:
            out.append(",")
    print(f"==== seg {k}: {len(segs.get(k, []))} instructions")
    print("".join(out))
