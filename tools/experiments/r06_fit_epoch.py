"""Round 6: where does fit() lose against the bare step?  Per-epoch wall time of fit(epochs=4) through callbacks, for 200 and
1000 batches per epoch, verbose 1 (device metrics) and 0."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from satrans_amd.callbacks import Callback  # noqa: E402

B = 8192


class Clock(Callback):
    def __init__(self):
        self.t = []

    def on_epoch_begin(self, epoch, logs=None):
        self.t.append(("begin", time.perf_counter()))

    def on_epoch_end(self, epoch, logs=None):
        self.t.append(("end", time.perf_counter()))


model = bench.build_model("cpu", 0.005)
model.to("cuda:0"); model.device = "cuda:0"
for nb in (200, 1000):
    X, y = bench.synth_batches(nb * B, 777)
    x = {f: X[:, i].astype(np.int64) for i, f in enumerate(bench.ALICCP_FIELDS)}
    for verbose in (1, 0):
        c = Clock()
        devnull = open(os.devnull, "w")
        old = sys.stdout, sys.stderr
        sys.stdout = sys.stderr = devnull
        try:
            model.fit(x=x, y=y, batch_size=B, epochs=4, verbose=verbose, shuffle=True, callbacks=[c])
        finally:
            sys.stdout, sys.stderr = old
        torch.cuda.synchronize()
        ep = [(c.t[2 * i + 1][1] - c.t[2 * i][1]) for i in range(4)]
        gap = [(c.t[2 * i + 2][1] - c.t[2 * i + 1][1]) for i in range(3)]
        print(f"{nb} batches, verbose={verbose}: epoch begin->end ms {[round(1e3 * v, 1) for v in ep]} = per step "
              f"{[round(1e3 * v / nb, 4) for v in ep]}; end->next begin ms {[round(1e3 * v, 2) for v in gap]}")
