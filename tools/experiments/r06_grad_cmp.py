import torch, sys
a=torch.load(sys.argv[1]); b=torch.load(sys.argv[2])
print("bce", float(a["_bce"]), float(b["_bce"]))
worst=[]
for k in a:
    if k.startswith("_"): continue
    d=(a[k]-b[k]).abs().max().item(); s=max(a[k].abs().max().item(),1e-30)
    worst.append((d/s,k,d,s))
worst.sort(reverse=True)
for w in worst[:12]: print(f"{w[0]:.3e}  {w[1]:60s} maxdiff {w[2]:.3e} scale {w[3]:.3e}")
