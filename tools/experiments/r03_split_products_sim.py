"""CPU simulation of bf16-split matrix products (VERDICT r02 item 1's "bf16 x 3" experiment): what do the logits and the gradients of
the AliCCP-shaped golden case lose when every weight product of the layer stack (projections, MetaNet, Out_linear; forward AND
backward) is evaluated as  a_hi w_hi + a_hi w_lo + a_lo w_hi  (x3) or with the a_lo w_lo term too (x4), hi = bf16(v),
lo = bf16(v - hi), products exact, accumulation in fp64 here (the MFMA accumulates in fp32: this is the optimistic side)?
Attention products stay fp32 (they are VALU work in the kernels).  Test infrastructure: uses the oracle.

    python tools/experiments/r03_split_products_sim.py [case]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import satrans_oracle as O  # noqa: E402
from tests.helpers import Case  # noqa: E402

TERMS = 3
MM = torch.matmul          # the real one (run() patches torch.matmul)


def bf(v):
    return v.to(torch.bfloat16).to(torch.float32)


def split(v):
    hi = bf(v)
    return hi.double(), bf(v - hi).double()


def split_mm(a, w):
    ah, al = split(a)
    wh, wl = split(w)
    out = MM(ah, wh) + MM(ah, wl) + MM(al, wh)
    if TERMS == 4:
        out = out + MM(al, wl)
    return out.float()


class SplitMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, w):
        ctx.save_for_backward(a, w)
        return split_mm(a, w)

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        ga = split_mm(g, w.transpose(-1, -2))
        gw = split_mm(a.transpose(-1, -2), g)          # (the kernels keep these token contractions in fp32 MFMA for now)
        while gw.dim() > w.dim():
            gw = gw.sum(0)
        return ga, gw


def run(case, use_split, dtype=torch.float32):
    spec = case.spec()
    P = O.make_leaves(case.tensors("param", dtype))
    X, y = case.X, case.y.to(dtype)
    real_matmul, real_tmm = torch.matmul, torch.Tensor.__matmul__
    if use_split:
        # only the weight products: operands whose second factor is a parameter / generated weight (2-D or [B, u, u'])
        def mm(a, w):
            if a.dim() == 4 or w.dim() == 4:            # attention products [B, H, F, d]
                return real_matmul(a, w)
            return SplitMM.apply(a, w)
        torch.matmul = mm
        torch.Tensor.__matmul__ = lambda self, other: mm(self, other)
    try:
        x = O.gather_fields(P, X, spec)
        vecs = O.scenario_vectors(P, X, spec)
        h = x
        for l in range(spec.layer_num):
            h = O.layer_forward(P, l, h, vecs[l], spec, O.Dropper("off"))
    finally:
        torch.matmul, torch.Tensor.__matmul__ = real_matmul, real_tmm
    logit = torch.nn.functional.linear(h.flatten(1), P["dnn_linear.weight"], P["dnn_linear.bias"])
    O.bce_sum(torch.sigmoid(logit), y).backward()
    return logit.detach(), {k: t.grad.clone() for k, t in P.items() if t.grad is not None}


def main():
    global TERMS
    name = sys.argv[1] if len(sys.argv) > 1 else "aliccp_sota"
    case = Case(name)
    ref_logit, ref_g = run(case, False, torch.float64)            # the yardstick: the same graph in fp64
    for terms in (0, 3, 4):                                        # 0: plain fp32 products
        TERMS = terms
        logit, g = run(case, terms != 0)
        worst = max(((g[k] - ref_g[k]).abs().max() / ref_g[k].abs().max().clamp_min(1e-30)).item() for k in ref_g
                    if ref_g[k].abs().max() > 0)
        which = max((k for k in ref_g if ref_g[k].abs().max() > 0),
                    key=lambda k: ((g[k] - ref_g[k]).abs().max() / ref_g[k].abs().max()).item())
        print(f"{name}: {('bf16 x%d' % terms) if terms else 'plain fp32'}: logit max abs err {float((logit - ref_logit).abs().max()):.3e} (|logit| max "
              f"{float(ref_logit.abs().max()):.2f}); worst gradient error / max|gradient| {worst:.3e} ({which})")


if __name__ == "__main__":
    main()
