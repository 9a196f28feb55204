"""Full-batch gradients of one training forward+backward (B = 8192, dropout on, fixed seeds) written to a file: run once per
library variant (SATRANS_LIB_PATH) and compare - python tools/experiments/r06_grad_dump.py out.pt [steps_before]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
cfg = bench.make_config("aliccp")
model = bench.build_model("cpu", cfg["lr"], cfg=cfg)
model.to("cuda:0"); model.device = "cuda:0"
eng = model._require_engine()
B = 8192
X, y = bench.synth_batches(4 * B, 5, cfg=cfg)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
model.train()
eng.drop_step = 40
bce, reg, grads = eng.loss_and_grads(Xd[:B], yd[:B])
out = {k: v.detach().cpu() for k, v in grads.items() if v.numel() < 2_000_000}
out["_bce"] = torch.tensor(bce)
torch.save(out, sys.argv[1])
print("bce", bce, "tensors", len(out))
