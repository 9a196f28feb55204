"""Wall time of `fit` through the public API at the BASELINE shape (full-size tables, N synthetic rows resident in HBM),
with the per-step train metrics of the reference's main.py (binary_crossentropy + auc, verbose > 0)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60 * 8192
model = bench.build_model("cpu", 0.005)
model.to("cuda:0"); model.device = "cuda:0"
X, y = bench.synth_batches(N, 5)
x = {f: X[:, i].astype(np.int64) for i, f in enumerate(bench.ALICCP_FIELDS)}
for verbose in (0, 2):
    torch.cuda.synchronize()
    t0 = time.time()
    model.fit(x=x, y=y, batch_size=8192, epochs=1, verbose=verbose, shuffle=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"fit verbose={verbose}: {N} rows in {dt:.3f}s = {N / dt / 1e6:.2f} M samples/s "
          f"(host metrics: {os.environ.get('SATRANS_HOST_METRICS', '0')})")
