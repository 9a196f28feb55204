"""Does a SECOND engine in the same process run as fast as the first?  (Round 5: it did not - its helper streams landed on the
hardware queues of the first engine's; the streams are shared per process since.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

B, n = 8192, 40
X, y = bench.synth_batches(n * B, 5)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
for k in range(4):
    model = bench.build_model("cpu", 0.005)
    model.to("cuda:0"); model.device = "cuda:0"
    eng = model._require_engine()
    model.train()
    for i in range(5):
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
    eng.flush_lazy(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5, n - 1):
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
    eng.flush_lazy(); torch.cuda.synchronize()
    print(f"engine {k}: {(time.perf_counter() - t0) / (n - 6) * 1e3:.4f} ms/step")
    del model, eng
