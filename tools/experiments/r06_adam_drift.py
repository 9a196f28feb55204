"""Round 6: the FAST arithmetic of the tables' Adam update (hardware v_sqrt_f32 / v_rcp_f32, satrans_adam_hparams.arith) against the
exact one (torch's operations bit for bit) - drift and cost.

  (1) kernel level, no feedback: one full-size AliCCP arena state (trained values: the bench model after 40 steps), then 700
      regulariser-only steps through the flush kernel in either arithmetic from the SAME state: max / median |p_fast - p_exact| / lr,
      moments compared bit for bit (they do not depend on the arithmetic of the update); flush ms per launch of 32 steps;
  (2) engine level, free running: 200 training steps (B = 8192, dropout on) in either arithmetic from one seed: quantiles of
      |p_fast - p_exact| / lr over the gathered tables (feedback through the gradients included), loss of both runs;
  (3) ms per step of `bench.py --train-only` style steps in either arithmetic on this box.
"""
import ctypes as C
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from satrans_amd import native as N  # noqa: E402

DEV = "cuda:0"
B = 8192
lib = N.lib()


def engine(arith):
    m = bench.build_model("cpu", 0.005)
    m.to(DEV)
    m.device = DEV
    e = m._require_engine()
    e.adam_arith = arith
    m.train()
    return m, e


def run(arith, steps, Xd, yd):
    m, e = engine(arith)
    n = Xd.shape[0] // B
    for i in range(5):
        e.train_step(Xd[(i % n) * B:(i % n + 1) * B], yd[(i % n) * B:(i % n + 1) * B], next_X=Xd[((i + 1) % n) * B:((i + 1) % n + 1) * B])
    e.flush_lazy()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(5, steps):
        e.train_step(Xd[(i % n) * B:(i % n + 1) * B], yd[(i % n) * B:(i % n + 1) * B], next_X=Xd[((i + 1) % n) * B:((i + 1) % n + 1) * B])
    e.flush_lazy()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (steps - 5) * 1e3
    return m, e, dt


X, y = bench.synth_batches(40 * B, 5)
Xd, yd = torch.from_numpy(X).to(DEV), torch.from_numpy(y).to(DEV)

# ---- (2) + (3) -------------------------------------------------------------------------------------------------------------
res = {}
for arith in ("exact", "fast", "exact", "fast"):
    m, e, dt = run(arith, 200, Xd, yd)
    res.setdefault(arith, []).append(dt)
    if len(res[arith]) == 1:
        res[arith + "_state"] = (m.embedding_arena.detach().clone(), e.adam_m.clone(), e.adam_v.clone(), float(e.epoch_sums()[0]))
    del m, e
print(f"(3) ms per step over 195 steps + their flushes: exact {res['exact']}, fast {res['fast']}")
pe, me, ve, le = res["exact_state"]
pf, mf, vf, lf = res["fast_state"]
d = ((pf - pe).abs() / 0.005).flatten()
moved = (pe != 0)
q = torch.quantile(d[::37].double(), torch.tensor([0.5, 0.99, 0.9999], dtype=torch.float64, device=DEV))
print(f"(2) free-running 200 steps: |p_fast - p_exact| / lr  median {float(q[0]):.3e}  99 % {float(q[1]):.3e}  99.99 % {float(q[2]):.3e}  "
      f"max {float(d.max()):.3e};  summed BCE exact {le:.4f} fast {lf:.4f} (rel {abs(lf - le) / le:.2e})")

# ---- (1) -------------------------------------------------------------------------------------------------------------------
R, D = pe.shape
lr, b1, b2, eps, l2 = 0.005, 0.9, 0.999, 1e-8, 1e-5
K, t0s = 700, 200
f32 = lambda x: float(np.float32(x))
table = torch.tensor([(0.0, 1.0)] * (t0s + 1) + [(f32(lr / (1.0 - b1 ** s)), 1.0 / f32(math.sqrt(1.0 - b2 ** s))) for s in range(t0s + 1, t0s + K + 1)],
                     dtype=torch.float64, device=DEV)
st = torch.cuda.current_stream().cuda_stream
out = {}
for arith in ("exact", "fast"):
    P, M, V = pe.clone(), me.clone(), ve.clone()
    last = torch.full((R,), t0s, dtype=torch.int32, device=DEV)
    regl = torch.zeros(int(lib.satrans_embed_lazy_reg_partials(64, D)), dtype=torch.float64, device=DEV)
    h = N.AdamHParams()
    h.lr_over_bc1, h.bc2_sqrt, h.beta1, h.beta2, h.eps, h.l2 = lr, 1.0, b1, b2, eps, l2
    h.arith = N.ADAM_FAST if arith == "fast" else N.ADAM_EXACT
    ms = []
    for target in range(t0s + 32, t0s + K + 1, 32):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        N.check(lib.satrans_embed_lazy_flush(P.data_ptr(), M.data_ptr(), V.data_ptr(), last.data_ptr(), R, D, target, table.data_ptr(),
                                             C.byref(h), 64, regl.data_ptr(), st), "flush")
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    out[arith] = (P, M, V, ms)
dP = ((out["fast"][0] - out["exact"][0]).abs() / lr).flatten()
print(f"(1) {len(out['exact'][3]) * 32} regulariser-only steps of {R:,} x {D} elements from one trained state: |p_fast - p_exact| / lr  "
      f"median {float(dP[::37].median()):.3e}  max {float(dP.max()):.3e};  moments identical: "
      f"{torch.equal(out['fast'][1], out['exact'][1]) and torch.equal(out['fast'][2], out['exact'][2])}")
print(f"    flush of 32 steps, ms per launch: exact first {out['exact'][3][0]:.2f} last {out['exact'][3][-1]:.2f} mean {np.mean(out['exact'][3]):.2f};  "
      f"fast first {out['fast'][3][0]:.2f} last {out['fast'][3][-1]:.2f} mean {np.mean(out['fast'][3]):.2f}")
