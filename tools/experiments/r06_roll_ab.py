"""A/B of the rolling flush forms on one box: python tools/experiments/r06_roll_ab.py [config] [steps] - pipelined steps per arm with the
periodic flushes inside and one flush at the end, three rounds; arms: "" (periodic flush), "tail", "step" (engine.rolling_flush)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
cfg = bench.make_config(sys.argv[1] if len(sys.argv) > 1 else "aliccp")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
bench.CFG = cfg
model = bench.build_model("cpu", cfg["lr"], cfg=cfg)
model.to("cuda:0"); model.device = "cuda:0"
eng = model._require_engine()
B, n = 8192, 16 if cfg.get("name") == "c5" or len(cfg["fields"]) > 32 else 64
X, y = bench.synth_batches(n * B, 5, cfg=cfg)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
model.train()


def run(steps, mode):
    eng.rolling_flush = mode
    for i in range(3):
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
    eng.flush_lazy(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        i = 3 + k % (n - 4)
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
    eng.flush_lazy(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for r in range(3):
    for mode in ("", "tail", "step"):
        print(f"round {r} rolling_flush={mode!r:7}: {run(steps, mode):.4f} ms/step", flush=True)
