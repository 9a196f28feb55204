import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
cfg = bench.make_config("aliccp")
model = bench.build_model("cpu", cfg["lr"], cfg=cfg)
model.to("cuda:0"); model.device = "cuda:0"
eng = model._require_engine()
B = 8192
X, y = bench.synth_batches(8 * B, 5, cfg=cfg)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
for phase in ("init", "trained"):
    if phase == "trained":
        model.train()
        for i in range(300):
            k = i % 7
            eng.train_step(Xd[k * B:(k + 1) * B], yd[k * B:(k + 1) * B], Xd[(k + 1) * B:(k + 2) * B])
    model.eval()
    for nb in (2048, 32768):
        model.set_forward_precision("fp32")
        model(Xd[:nb]); l32 = eng.last_logit().clone()
        model.set_forward_precision("bf16")
        res = {}
        for stack in (True, False):
            eng.bf16_stack = stack
            model(Xd[:nb]); res[stack] = eng.last_logit().clone()
        print(phase, nb, "stack==layers:", torch.equal(res[True], res[False]), "max|stack-layers|", float((res[True] - res[False]).abs().max()),
              "bf16 vs fp32:", float((res[False] - l32).abs().max()), "max|logit|", float(l32.abs().max()), flush=True)
    model.set_forward_precision("fp32")
