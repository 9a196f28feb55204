"""Round 6: the training step WITHOUT the forward-to-backward hand-over (engine.save_attention = False), for rocprofv3 --stats: what
the hand-over's stores cost the forward kernel and what its loads save the backward one, kernel by kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
model = bench.build_model("cpu", 0.005)
model.to("cuda:0"); model.device = "cuda:0"
eng = model._require_engine()
eng.save_attention = len(sys.argv) > 1 and sys.argv[1] == "save"
B, n = 8192, 34
X, y = bench.synth_batches(n * B, 5)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
model.train()
for i in range(n - 1):
    eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
eng.flush_lazy(); torch.cuda.synchronize()
