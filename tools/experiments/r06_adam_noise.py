"""Round 6: is the free-running difference between the fast and the exact Adam arithmetic larger than the difference between two
EXACT fp32 implementations?  200 training steps (B = 8192, dropout on, one seed) per process; prints the summed BCE after every
20 steps.  Run once per (arith, library):   python r06_adam_noise.py exact|fast   [SATRANS_LIB_PATH=another build of the kernels,
e.g. the wavefront attention arm - the same mathematics in another summation order]."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

arith = sys.argv[1]
B, steps = 8192, 200
m = bench.build_model("cpu", 0.005)
m.to("cuda:0"); m.device = "cuda:0"
e = m._require_engine()
e.adam_arith = arith
m.train()
X, y = bench.synth_batches(40 * B, 5)
Xd, yd = torch.from_numpy(X).to("cuda:0"), torch.from_numpy(y).to("cuda:0")
n = 40
out = []
for i in range(steps):
    e.train_step(Xd[(i % n) * B:(i % n + 1) * B], yd[(i % n) * B:(i % n + 1) * B])
    if i % 20 == 19:
        out.append(round(float(e.epoch_sums()[0]), 3))
e.flush_lazy()
p = m.embedding_arena.detach()
print(arith, os.environ.get("SATRANS_LIB_PATH", "shipped").split("/")[-1], "summed BCE:", out, "| sum |p| %.6f" % float(p.abs().sum()))
