"""900 pipelined steps for a rocprofv3 --kernel-trace; r05_trace900_report.py prints the flush durations and first / last quarter means."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
cfg = bench.make_config("aliccp")
bench.CFG = cfg
B, n = 8192, 64
X, y = bench.synth_batches(n * B, 5, cfg=cfg)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
model = bench.build_model("cpu", cfg["lr"], cfg=cfg)
model.to("cuda:0"); model.device = "cuda:0"
model.train()
eng = model._require_engine()
for k in range(900):
    i = k % (n - 1)
    eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
eng.flush_lazy(); torch.cuda.synchronize()
