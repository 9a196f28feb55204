"""A/B of an engine attribute on one box: python tools/experiments/r05_attr_ab.py <attribute> [config] - 256 pipelined steps per arm,
three rounds, flushes inside."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
attr = sys.argv[1]
cfg = bench.make_config(sys.argv[2] if len(sys.argv) > 2 else "aliccp")
bench.CFG = cfg
model = bench.build_model("cpu", cfg["lr"], cfg=cfg)
model.to("cuda:0"); model.device = "cuda:0"
eng = model._require_engine()
B, n = 8192, 64
X, y = bench.synth_batches(n * B, 5, cfg=cfg)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
model.train()


def run(steps, on):
    setattr(eng, attr, on)
    for i in range(5):
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
    eng.flush_lazy(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        i = 5 + k % (n - 6)
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
    eng.flush_lazy(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for r in range(3):
    for on in (False, True):
        print(f"round {r} {attr}={on}: {run(256, on):.4f} ms/step")
