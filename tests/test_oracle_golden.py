"""Pin the CPU oracle against vectors recorded from the reference itself (oracle/gen_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import satrans_oracle as O
from tests.helpers import ADAM_CASES, ALL_CASES, TRAIN_CASES, Case


@pytest.mark.parametrize("name", ALL_CASES)
def test_forward_matches_reference(name):
    c = Case(name)
    P, spec = c.tensors("param"), c.spec()
    trace = {}
    prob, logit = O.forward(P, c.X, spec, trace=trace)
    want = c.arrays("out")
    np.testing.assert_allclose(prob.numpy(), want["prob"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(logit.numpy(), want["logit"], rtol=0, atol=1e-6)
    assert np.array_equal(trace["att_input"].numpy(), want["att_input"])          # integer gather: bit-exact
    np.testing.assert_allclose(trace["vec0"].numpy()[:8], want["vec0"], rtol=0, atol=1e-7)
    for l in range(spec.layer_num):
        np.testing.assert_allclose(trace[f"out{l}"].numpy(), want[f"layer{l}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(trace[f"att{l}"].numpy(), want[f"att{l}"], rtol=0, atol=1e-6)
        for r in "qk":
            if f"{r}{l}" in want:
                np.testing.assert_allclose(trace[f"{r}{l}"].numpy(), want[f"{r}{l}"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_loss_and_grads_match_reference(name):
    c = Case(name)
    P, spec = c.tensors("param"), c.spec()
    bce, reg, grads = O.loss_and_grads(P, c.X, c.y, spec)
    z = c.z
    assert bce == pytest.approx(float(z["train/bce"]), rel=1e-6)
    assert reg == pytest.approx(float(z["train/reg"]), rel=1e-5)
    want = c.arrays("grad")
    assert set(want) <= set(grads), set(want) - set(grads)
    for k, g in want.items():
        scale = max(1e-6, float(np.abs(g).max()))
        np.testing.assert_allclose(grads[k].numpy(), g, rtol=0, atol=2e-5 * scale, err_msg=k)
    # keys the reference never gives a gradient stay without one
    for k in grads:
        assert k in want or any(torch.equal(grads[k], torch.from_numpy(want[a])) for a in want if want[a].shape == grads[k].shape), k


@pytest.mark.parametrize("name", ADAM_CASES)
def test_adam_steps_match_reference(name):
    c = Case(name)
    spec = c.spec()
    tr = O.OracleTrainer(c.tensors("param"), spec, lr=c.meta["lr"])
    for _ in range(c.meta["adam_steps"]):
        tr.step(c.X, c.y)
    got, want = tr.state(), c.tensors("adam")
    grads = c.arrays("grad")
    for k, w in want.items():
        scale = max(1e-6, float(w.abs().max()))
        atol = 3e-5 * scale
        if k in grads and float(np.abs(grads[k]).max()) < 1e-7:
            # a mathematically-zero gradient (e.g. the K-side MetaNet LayerNorm bias: softmax is invariant to a
            # constant added to every key) is rounding noise that Adam's g/(sqrt(v)+eps) amplifies
            atol = 0.05 * c.meta["lr"]
        np.testing.assert_allclose(got[k].numpy(), w.numpy(), rtol=0, atol=atol, err_msg=k)
    # torch.optim.Adam's moments on the reference's trajectory: these are well-conditioned (linear / quadratic in the
    # gradients), so they pin the optimizer far more tightly than the parameters can
    for kind in ("exp_avg", "exp_avg_sq"):
        for k, w in c.arrays(f"opt/{kind}").items():
            if k in grads and float(np.abs(grads[k]).max()) < 1e-7:
                continue                                   # mathematically-zero gradient: its moments are rounding noise
            st = tr.optim.state[tr.leaves[k]][kind].numpy()
            scale = float(np.abs(w).max())
            np.testing.assert_allclose(st, w, rtol=1e-5, atol=(2e-5 if kind == "exp_avg" else 1e-4) * scale + 1e-30,
                                       err_msg=f"{kind}/{k}")


def test_fp64_oracle_agrees_with_fp32():
    """The oracle is dtype-generic; fp64 is the yardstick for the fp32 tolerance."""
    c = Case("aliccp_sota")
    spec = c.spec()
    p32, l32 = O.forward(c.tensors("param"), c.X, spec)
    p64, l64 = O.forward(c.tensors("param", torch.float64), c.X.double(), spec)
    assert float((l32.double() - l64).abs().max()) < 5e-6
