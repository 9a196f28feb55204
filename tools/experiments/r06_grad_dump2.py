"""Gradient agreement of two library builds on a TRAINED state: `make` trains 300 steps and saves the state_dict and the gradients of a
probe batch; `check` loads that state_dict in a process that runs another library (SATRANS_LIB_PATH) and compares its gradients.
    python tools/experiments/r06_grad_dump2.py make /tmp/s.pt ;  SATRANS_LIB_PATH=... python tools/experiments/r06_grad_dump2.py check /tmp/s.pt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
mode, path = sys.argv[1], sys.argv[2]
cfg = bench.make_config("aliccp")
model = bench.build_model("cpu", cfg["lr"], cfg=cfg)
B = 8192
X, y = bench.synth_batches(9 * B, 5, cfg=cfg)
if mode == "check":
    saved = torch.load(path)
    model.load_state_dict(saved["state"])
model.to("cuda:0"); model.device = "cuda:0"
eng = model._require_engine()
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
model.train()
if mode == "make":
    for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 300):
        k = i % 7
        eng.train_step(Xd[k * B:(k + 1) * B], yd[k * B:(k + 1) * B], Xd[(k + 1) * B:(k + 2) * B])
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
eng.drop_step = 1234
bce, reg, grads = eng.loss_and_grads(Xd[8 * B:9 * B], yd[8 * B:9 * B])
g = {k: v.detach().cpu() for k, v in grads.items() if v.numel() < 2_000_000}
logit = eng._ws[B]["logit"].detach().cpu()
if mode == "make":
    torch.save({"state": state, "grads": g, "bce": bce}, path)
    print("made: bce", bce, "max|logit|", float(logit.abs().max()))
else:
    print("bce", saved["bce"], bce, "max|logit|", float(logit.abs().max()))
    worst = []
    for k in g:
        d = (g[k] - saved["grads"][k]).abs().max().item(); s = max(saved["grads"][k].abs().max().item(), 1e-30)
        worst.append((d / s, k, d, s))
    worst.sort(reverse=True)
    for w in worst[:8]:
        print(f"{w[0]:.3e}  {w[1]:56s} maxdiff {w[2]:.3e} scale {w[3]:.3e}")
