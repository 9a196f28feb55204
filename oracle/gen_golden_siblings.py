"""Golden vectors for the sibling users of the attention-path kernels, recorded from the REFERENCE's own
classes in the build container (same rules as oracle/gen_golden.py: the reference is imported, never copied; the fixtures are data).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_siblings.py        # writes tests/golden/sibling_*.npz

  sibling_selfatt_*: models/submodules.py SelfAttention_Layer - seeded parameters, input block, eval output, attention scores,
                     and (dropout p = 0, train mode) the gradients of sum(y * w) with respect to the input and every parameter
  sibling_metanet_*: the call sequence of BaseModel.meta_transformation (models/basemodel.py:191-199) on the reference's
                     DNN_v2 + MetaNet + nn.Embedding: scenario ids, input block, output, gradients of the same functional
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(ROOT, "oracle", "shims"), ROOT, "/root/reference"]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from models.submodules import DNN_v2, MetaNet, SelfAttention_Layer  # noqa: E402  (the reference)


def zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0


def selfatt_case(name, D, H, Fn, B, use_res, scaling, outdir):
    torch.manual_seed(7)
    layer = SelfAttention_Layer(D, head_num=H, use_res=use_res, scaling=scaling)
    rng = np.random.RandomState(3)
    x = torch.from_numpy(rng.randn(B, Fn, D).astype(np.float32) * 0.5)
    w = torch.from_numpy(rng.randn(B, Fn, D).astype(np.float32))
    out = {f"param/{k}": v.detach().numpy().copy() for k, v in layer.state_dict().items()}
    layer.eval()
    with torch.no_grad():
        y = layer(x.clone())
    out["x"], out["w"], out["y"] = x.numpy(), w.numpy(), y.numpy().copy()
    out["att"] = layer.normalized_att_scores.detach().numpy().copy()
    zero_dropout(layer)
    layer.train()
    xg = x.clone().requires_grad_(True)
    (layer(xg) * w).sum().backward()
    out["grad/x"] = xg.grad.numpy().copy()
    for k, p in layer.named_parameters():
        if p.grad is not None:
            out[f"grad/{k}"] = p.grad.numpy().copy()
    out["meta"] = np.array(repr(dict(D=D, H=H, F=Fn, B=B, use_res=use_res, scaling=scaling)))
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    print(name, len(out), "arrays")


def metanet_case(name, D, U, Fn, B, S, use_norm, outdir):
    torch.manual_seed(11)
    emb = nn.Embedding(S, D)                                        # models/basemodel.py:139
    P = D * U + U * D
    enc = DNN_v2(D, [P])                                            # :144
    net = MetaNet(hidden_dim=D, use_norm=use_norm, meta_dnn_hidden_units=(D, U, D))     # :145-148
    if use_norm:
        with torch.no_grad():                                       # make the norm's parameters non-trivial
            net.ffn_layer_norm.weight.normal_(1.0, 0.2)
            net.ffn_layer_norm.bias.normal_(0.0, 0.2)
    with torch.no_grad():
        enc.linears[0].weight.normal_(0.0, 0.05)                    # (1e-4 would make every product vanish in fp32 noise)
    rng = np.random.RandomState(5)
    ids = torch.from_numpy(rng.randint(0, S, size=B).astype(np.int64))
    ids[0] = S - 1
    x = torch.from_numpy(rng.randn(B, Fn, D).astype(np.float32) * 0.5)
    w = torch.from_numpy(rng.randn(B, Fn, D).astype(np.float32))

    def run(xin):                                                   # models/basemodel.py:191-199
        domain_emb = F.relu(emb(ids))
        domain_vec = enc(domain_emb)
        return net(xin, domain_vec)

    out = {"param/domain_embeddings.weight": emb.weight.detach().numpy().copy(),
           "param/domain_map_dnn.weight": enc.linears[0].weight.detach().numpy().copy(),
           "param/domain_map_dnn.bias": enc.linears[0].bias.detach().numpy().copy()}
    if use_norm:
        out["param/ffn_layer_norm.weight"] = net.ffn_layer_norm.weight.detach().numpy().copy()
        out["param/ffn_layer_norm.bias"] = net.ffn_layer_norm.bias.detach().numpy().copy()
    net.eval()
    with torch.no_grad():
        y = run(x.clone())
    out["ids"], out["x"], out["w"], out["y"] = ids.numpy(), x.numpy(), w.numpy(), y.numpy().copy()
    zero_dropout(net)
    net.train()
    xg = x.clone().requires_grad_(True)
    (run(xg) * w).sum().backward()
    out["grad/x"] = xg.grad.numpy().copy()
    out["grad/domain_embeddings.weight"] = emb.weight.grad.numpy().copy()
    out["grad/domain_map_dnn.weight"] = enc.linears[0].weight.grad.numpy().copy()
    out["grad/domain_map_dnn.bias"] = enc.linears[0].bias.grad.numpy().copy()
    if use_norm:
        out["grad/ffn_layer_norm.weight"] = net.ffn_layer_norm.weight.grad.numpy().copy()
        out["grad/ffn_layer_norm.bias"] = net.ffn_layer_norm.bias.grad.numpy().copy()
    out["meta"] = np.array(repr(dict(D=D, U=U, F=Fn, B=B, S=S, use_norm=use_norm)))
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
    print(name, len(out), "arrays")


if __name__ == "__main__":
    outdir = os.path.join(ROOT, "tests", "golden")
    selfatt_case("sibling_selfatt_d32", 32, 4, 19, 24, True, True, outdir)
    selfatt_case("sibling_selfatt_d16_nores", 16, 2, 7, 9, False, False, outdir)
    selfatt_case("sibling_selfatt_d64", 64, 4, 40, 6, True, True, outdir)
    metanet_case("sibling_metanet_d32", 32, 64, 19, 24, 4, False, outdir)
    metanet_case("sibling_metanet_d16_norm", 16, 32, 7, 9, 3, True, outdir)
