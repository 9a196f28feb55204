"""Diagnostic: where `fit` spends its host time (cProfile over one epoch of 200 batches at the BASELINE shape)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

model = bench.build_model("cpu", 0.005)
model.to("cuda:0"); model.device = "cuda:0"
B, nb = 8192, 200
X, y = bench.synth_batches(nb * B, 777)
x = {f: X[:, i].astype(np.int64) for i, f in enumerate(bench.ALICCP_FIELDS)}
model.fit(x={k: v[:4 * B] for k, v in x.items()}, y=y[:4 * B], batch_size=B, epochs=1, verbose=0, shuffle=True)   # warm
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
model.fit(x=x, y=y, batch_size=B, epochs=1, verbose=int(sys.argv[1]) if len(sys.argv) > 1 else 1, shuffle=True)
torch.cuda.synchronize()
pr.disable()
print(f"fit wall {time.perf_counter() - t0:.3f}s for {nb} steps", file=sys.stderr)
pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(28)
