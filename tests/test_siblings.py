"""Sibling users of the attention-path kernels: reference SelfAttention_Layer and
BaseModel.meta_transformation.  CPU: the oracle restatements against vectors recorded from the reference's own classes
(oracle/gen_golden_siblings.py).  GPU: satrans_amd.layers (HIP launches behind torch.autograd.Function) against the same
vectors, forward and every gradient."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import satrans_oracle as O
from tests.helpers import GOLDEN, SIBLING_METANET, SIBLING_SELFATT


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = ast.literal_eval(str(z["meta"]))
    P = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    G = {k[5:]: z[k] for k in z.files if k.startswith("grad/")}
    return z, meta, P, G


def close(got, want, rel, msg):
    scale = max(1e-6, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=0, atol=rel * scale + 1e-9, err_msg=msg)


@pytest.mark.parametrize("name", SIBLING_SELFATT)
def test_oracle_selfattention_matches_reference(name):
    z, m, P, G = load(name)
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    y, att = O.selfattention_layer(leaves, x, m["H"], m["use_res"], m["scaling"])
    close(y.detach().numpy(), z["y"], 2e-6, "y")
    close(att.detach().numpy(), z["att"], 2e-6, "att")
    (y * torch.from_numpy(z["w"])).sum().backward()
    close(x.grad.numpy(), G["x"], 2e-5, "grad x")
    for k, g in G.items():
        if k != "x":
            close(leaves[k].grad.numpy(), g, 2e-5, k)
    assert "W_Out" not in G                       # a parameter the reference never uses


@pytest.mark.parametrize("name", SIBLING_METANET)
def test_oracle_meta_transformation_matches_reference(name):
    z, m, P, G = load(name)
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    y = O.meta_transformation(leaves, torch.from_numpy(z["ids"]), x, [m["D"], m["U"], m["D"]], m["use_norm"])
    close(y.detach().numpy(), z["y"], 2e-6, "y")
    (y * torch.from_numpy(z["w"])).sum().backward()
    close(x.grad.numpy(), G["x"], 2e-5, "grad x")
    for k, g in G.items():
        if k != "x":
            close(leaves[k].grad.numpy(), g, 2e-5, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SIBLING_SELFATT)
def test_selfattention_layer_on_the_gpu_matches_reference(name):
    from satrans_amd import SelfAttention_Layer
    z, m, P, G = load(name)
    torch.manual_seed(7)                          # the reference's creation order and init: same seed, same parameters
    layer = SelfAttention_Layer(m["D"], head_num=m["H"], use_res=m["use_res"], scaling=m["scaling"])
    for k, v in layer.state_dict().items():
        assert torch.equal(v, P[k]), f"seeded parameter {k} differs from the reference's"
    layer.to("cuda:0")
    x = torch.from_numpy(z["x"]).to("cuda:0")
    layer.eval()
    layer.capture_attention = True
    y = layer(x)
    close(y.detach().cpu().numpy(), z["y"], 1e-5, "y")
    close(layer.normalized_att_scores.cpu().numpy(), z["att"], 2e-6, "att")
    layer.capture_attention = False
    xg = x.clone().requires_grad_(True)
    (layer(xg) * torch.from_numpy(z["w"]).to("cuda:0")).sum().backward()      # eval mode = dropout off, like the recorded step
    close(xg.grad.cpu().numpy(), G["x"], 5e-5, "grad x")
    grads = {k: p.grad for k, p in layer.named_parameters()}
    for k, g in G.items():
        if k != "x":
            close(grads[k].cpu().numpy(), g, 5e-5, k)
    assert grads["W_Out"] is None
    # training mode: dropout on, deterministic for a (seed, step), different from eval
    layer.train()
    y1 = layer(x)
    assert not torch.equal(y1, y) and torch.isfinite(y1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", SIBLING_METANET)
def test_meta_transformation_on_the_gpu_matches_reference(name):
    from satrans_amd import MetaTransformation
    z, m, P, G = load(name)
    mod = MetaTransformation(m["D"], m["S"] - 1, (m["D"], m["U"], m["D"]), use_norm=m["use_norm"])
    sd = {"domain_embeddings.weight": P["domain_embeddings.weight"], "domain_map_dnn.weight": P["domain_map_dnn.weight"],
          "domain_map_dnn.bias": P["domain_map_dnn.bias"]}
    if m["use_norm"]:
        sd["ffn_layer_norm.weight"], sd["ffn_layer_norm.bias"] = P["ffn_layer_norm.weight"], P["ffn_layer_norm.bias"]
    mod.load_state_dict(sd)
    mod.to("cuda:0")
    mod.eval()
    ids = torch.from_numpy(z["ids"]).to("cuda:0")
    x = torch.from_numpy(z["x"]).to("cuda:0")
    y = mod(ids, x)
    close(y.detach().cpu().numpy(), z["y"], 1e-5, "y")
    xg = x.clone().requires_grad_(True)
    (mod(ids, xg) * torch.from_numpy(z["w"]).to("cuda:0")).sum().backward()
    close(xg.grad.cpu().numpy(), G["x"], 5e-5, "grad x")
    grads = {k: p.grad for k, p in mod.named_parameters()}
    for k, g in G.items():
        if k != "x":
            close(grads[k].cpu().numpy(), g, 5e-5, k)


@pytest.mark.gpu
@pytest.mark.parametrize("D,U", [(16, 16), (16, 48), (32, 32), (32, 128), (64, 16), (64, 96), (64, 128), (128, 16), (128, 64)])
@pytest.mark.parametrize("use_norm", [True, False])
def test_meta_transformation_shape_sweep_against_the_oracle(D, U, use_norm):
    """The general-path products behind MetaTransformation (register-direct MFMA products for every K / 16, N / 16 in 1..8,
    weight-gradient products of the 1x1, 1x2 and 2x1 block forms) on shapes no golden case holds: forward and every gradient
    against the oracle's meta_transformation - itself pinned to the reference by the recorded vectors above - on a ragged
    batch with uneven, partly empty scenario segments."""
    from satrans_amd import MetaTransformation
    g = torch.Generator().manual_seed(1000 + 7 * D + U)
    S, B, F = 5, 37, 11
    mod = MetaTransformation(D, S - 1, (D, U, D), use_norm=use_norm)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.ndim > 1 else 0.1))
        if use_norm:
            mod.ffn_layer_norm.weight.copy_(1.0 + 0.2 * torch.randn(D, generator=g))
    P = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    ids = torch.tensor([0, 1, 3, 3, 1, 0, 3] * 6)[:B]                     # scenario 2 and 4 stay empty
    x = torch.randn(B, F, D, generator=g)
    w = torch.randn(B, F, D, generator=g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    y_ref = O.meta_transformation(leaves, ids, xr, [D, U, D], use_norm)
    (y_ref * w).sum().backward()
    mod.to("cuda:0")
    mod.eval()
    xg = x.to("cuda:0").requires_grad_(True)
    y = mod(ids.to("cuda:0"), xg)
    close(y.detach().cpu().numpy(), y_ref.detach().numpy(), 2e-5, "y")
    (y * w.to("cuda:0")).sum().backward()
    close(xg.grad.cpu().numpy(), xr.grad.numpy(), 1e-4, "grad x")
    for k, p in mod.named_parameters():
        want = leaves[k].grad
        assert want is not None, k
        close(p.grad.cpu().numpy(), want.numpy(), 1e-4, k)
