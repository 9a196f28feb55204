import os, sys, time, json
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/satrans_amd") else os.getcwd())
import torch
import bench
model = bench.build_model("cpu", 0.005)
model.to("cuda:0"); model.device = "cuda:0"
eng = model._require_engine()
B = 8192
n = 64
X, y = bench.synth_batches(n * B, 5)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
model.train()
def run(steps, ahead):
    eng.replay_ahead = ahead
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
    for ahead in (False, True):
        print(f"round {r} replay_ahead={ahead}: {run(256, ahead):.4f} ms/step (256 steps, flushes inside)")
