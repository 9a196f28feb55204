"""Diagnostic: compute-side cost of one rank's step in an N-rank job, measured on ONE GPU.

`parallel.world_size / gather_rows / gather_grad_rows_async / all_reduce_flat` are replaced by stand-ins that append the arena rows of N-1
other synthetic batches (what the all-gather of ids would deliver) and tile the local gradient rows N times (what
the all-gather of gradient rows would deliver; the tiling copy costs about what writing the received buffer does).
Everything downstream - sort, lazy replay, segmented sums, touched-row Adam over N*B*F rows - is the real code.
No communication time is included: add the all-gather estimate of DESIGN.md to read a scaling number from this.

    python tools/fake_world.py [N ...]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from satrans_amd import parallel  # noqa: E402


def run(N, steps=12, warmup=4, B=8192):
    model = bench.build_model("cpu", 0.005)
    model.to("cuda:0"); model.device = "cuda:0"
    eng = model._require_engine()
    model.train()
    X, y = bench.synth_batches((steps + warmup) * B, 100)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    others = []
    if N > 1:
        Xo, _ = bench.synth_batches((steps + warmup) * B * (N - 1), 777)
        Xo = torch.from_numpy(Xo).cuda().long()
        big = eng.row_span[:, 0] >= eng.small_rows                                  # large-table fields only
        off = eng.row_span[big, 0][None, :]
        rows_o = (Xo[:, eng.cols.long()[big]] + off).to(torch.int32).reshape(steps + warmup, -1)
        others = [rows_o[i].contiguous() for i in range(steps + warmup)]
    state = {"i": 0}
    parallel.world_size = lambda: N
    parallel.exchange_enabled = lambda: N > 1          # the step's multi-rank branch, fed by the stand-ins below
    parallel.gather_rows = lambda rows: torch.cat([rows.reshape(-1), others[state["i"]]]) if N > 1 else rows.reshape(-1)
    parallel.gather_grad_rows_async = lambda g: ((g.reshape(-1, g.shape[-1]).repeat(N, 1) if N > 1 else g.reshape(-1, g.shape[-1])), None)
    parallel.all_reduce_flat = lambda flat: None
    eng.timers = None
    for i in range(warmup):
        state["i"] = i
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        state["i"] = i
        eng.timers = {} if i == warmup + steps - 1 else None
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B])
    ph = eng.phase_ms()
    eng.timers = None
    eng.flush_lazy()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"N={N}: {ms:.3f} ms/step per rank (no communication) -> {N * B / ms / 1e3:.2f} M samples/s if comm were free;"
          f" phases of the last step: " + ", ".join(f"{k} {v:.3f}" for k, v in ph.items()))


if __name__ == "__main__":
    for n in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]):
        run(n)
