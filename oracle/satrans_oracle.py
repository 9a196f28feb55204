"""CPU oracle for the SATrans hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the shipped path (`satrans_amd/`) never does and fails
loudly when the HIP library is missing.

What this is: a from-scratch, op-for-op restatement (torch CPU, fp32 or fp64)
of the reference's forward/loss/regulariser/optimizer step for the path
SURVEY.md §8(a) lists.  It is written functionally over a flat dict of tensors
that uses the reference's `state_dict()` key names, so a state_dict captured
from the reference (tests/golden/*.npz) or one taken from the product model can
be evaluated by the very same code.

Parity pinning: the reference holds no tests or golden vectors of its own
(SURVEY.md §4), so this oracle is pinned against outputs of the reference
itself, generated in the build container by `oracle/gen_golden.py` (which
imports /root/reference with shims for its four uninstalled third-party
packages) and committed under `tests/golden/`.  `tests/test_oracle_golden.py`
replays every fixture through this file.

Reference lines each function follows are given in its docstring
(paths relative to the reference root).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# Static description of one model instance (shapes + flags), independent of any package.
# --------------------------------------------------------------------------------------
@dataclass
class PathSpec:
    """Everything `forward` needs besides the tensors.

    sparse      : [(embedding key name, X column)] in dnn_feature_columns order
    dense       : [(X col start, X col end)] of the DenseFeat columns, in order
    domain_cols : X column of every scenario column (first one drives the MetaNet)
    """
    sparse: List[Tuple[str, int]]
    dense: List[Tuple[int, int]]
    domain_cols: List[int]
    embedding_dim: int
    head_num: int
    layer_num: int
    flag: str = "sota"
    meta_mode: str = "QK"
    meta_units: Sequence[int] = (32, 64, 32)      # [D] + meta_dnn_hidden_units
    l2_reg_embedding: float = 1e-5
    use_res: bool = True
    drop_rate: float = 0.1                         # models/satrans.py:27-28
    multi_domain_sparse: List[Tuple[str, int]] = field(default_factory=list)

    @property
    def meta_param_size(self) -> int:              # models/satrans.py:30
        u = list(self.meta_units)
        return sum(u[i] * u[i + 1] for i in range(len(u) - 1))


class Dropper:
    """The four dropout sites of a layer (MetaNet-Q, MetaNet-K, attention probabilities,
    output projection; models/submodules.py:97, models/satrans.py:87,94).

    mode 'off'   : evaluation, identity
    mode 'torch' : torch's own CPU generator (what the reference does; used for the CPU baseline)
    mode 'masks' : caller-supplied multiplicative masks (already scaled by 1/(1-p)), keyed by
                   (layer, site); lets a test replay the HIP kernels' counter-based masks exactly
    """

    def __init__(self, mode: str = "off", p: float = 0.1, masks: Optional[Dict] = None):
        assert mode in ("off", "torch", "masks")
        self.mode, self.p, self.masks = mode, p, masks or {}

    def __call__(self, x: Tensor, layer: int, site: str) -> Tensor:
        if self.mode == "off" or self.p == 0.0:
            return x
        if self.mode == "torch":
            return F.dropout(x, self.p, training=True)
        return x * self.masks[(layer, site)].to(x.dtype)


# --------------------------------------------------------------------------------------
# Forward pieces
# --------------------------------------------------------------------------------------
def gather_fields(P: Dict[str, Tensor], X: Tensor, spec: PathSpec, prefix: str = "embedding_dict") -> Tensor:
    """Per-field row lookup and concat to [B, F, D].

    models/meta_basemodel.py:533-535 (`emb[name](X[:, s:e].long())` per SparseFeat) followed by
    `concat_fun(..., axis=1)` at models/satrans.py:211.  Ids arrive as floats and are truncated by
    `.long()`.
    """
    rows = []
    for name, col in spec.sparse:
        ids = X[:, col].long()
        rows.append(P[f"{prefix}.{name}.weight"][ids])
    return torch.stack(rows, dim=1)


def dense_block(X: Tensor, spec: PathSpec) -> Optional[Tensor]:
    """models/meta_basemodel.py:542-543 + models/satrans.py:247-249."""
    if not spec.dense:
        return None
    return torch.cat([X[:, s:e] for s, e in spec.dense], dim=1)


def scenario_embedding(P: Dict[str, Tensor], X: Tensor, spec: PathSpec) -> Tensor:
    """relu(domain_embeddings[ids]); models/satrans.py:203-207,213."""
    ids = X[:, spec.domain_cols[0]].long()
    emb = P["domain_embeddings.weight"][ids]
    if len(spec.domain_cols) > 1:
        # models/satrans.py:205-207: mean over the scenario columns' rows of a second table set
        stack = [P[f"domain_embedding_dict.{name}.weight"][X[:, col].long()] for name, col in spec.multi_domain_sparse]
        emb = torch.stack(stack, dim=-1).mean(-1)
    return torch.relu(emb)


def scenario_encoder(P: Dict[str, Tensor], z: Tensor, spec: PathSpec) -> Tensor:
    """`domain_map_dnn_Q`: one Linear(in -> P) with bias and no activation
    (models/submodules.py:47-61 with hidden_units=[P]; built at models/satrans.py:173-180).
    With flag 'onlyemb' it is the identity."""
    if "onlyemb" in spec.flag:
        return z
    return F.linear(z, P["domain_map_dnn_Q.linears.0.weight"], P["domain_map_dnn_Q.linears.0.bias"])


def scenario_vectors(P: Dict[str, Tensor], X: Tensor, spec: PathSpec) -> List[List[Tensor]]:
    """Per layer, the three generated-weight vectors [vec_Q, vec_K, vec_V], each [B, >=P].

    models/satrans.py:217-234.  Without 'pos' one vector serves every layer and role; with 'pos'
    the encoder input is relu(cat[relu(dom_emb), layerid_emb[l] + qkvid_emb[r]]).
    """
    dom = scenario_embedding(P, X, spec)
    out: List[List[Tensor]] = []
    if "pos" not in spec.flag:
        vec = scenario_encoder(P, dom, spec)
        return [[vec, vec, vec] for _ in range(spec.layer_num)]
    B = X.shape[0]
    for l in range(spec.layer_num):
        lay = P["layerid_embeddings.weight"][l].expand(B, -1)
        trio = []
        for r in range(3):
            role = P["qkvid_embeddings.weight"][r].expand(B, -1)
            z = torch.relu(torch.cat([dom, lay + role], dim=1))
            trio.append(scenario_encoder(P, z, spec))
        out.append(trio)
    return out


# Test instrumentation: when a dict {"eps": e, "near_zero": 0} is installed here, metanet() counts the hidden units whose
# pre-activation lies within e * max|pre-activation| of the ReLU's kink.  Such a unit may legitimately take the other branch
# in another fp32 evaluation order; the parity tests allow their kink exception only when this count is non-zero.
KINK_PROBE: Optional[Dict] = None


def metanet(x: Tensor, vec: Tensor, gamma: Tensor, beta: Tensor, spec: PathSpec,
            drop: Dropper, layer: int, site: str) -> Tensor:
    """Per-sample generated MLP, no biases: LN(dropout(relu(x@W1)@W2...) + x).

    models/submodules.py:77-103.  `vec[:, off:off+u_i*u_{i+1}]` is reshaped row-major to
    [B, u_i, u_{i+1}]; ReLU after every matrix except the last; eps = 1e-6.
    """
    units = list(spec.meta_units)
    res = x
    off = 0
    for i in range(len(units) - 1):
        n = units[i] * units[i + 1]
        W = vec[:, off:off + n].reshape(-1, units[i], units[i + 1])
        off += n
        x = torch.matmul(x, W)
        if i < len(units) - 2:
            if KINK_PROBE is not None:
                xa = x.detach().abs()
                KINK_PROBE["near_zero"] += int((xa <= KINK_PROBE["eps"] * float(xa.max())).sum())
            x = torch.relu(x)
    x = drop(x, layer, site) + res
    return F.layer_norm(x, (x.shape[-1],), gamma, beta, 1e-6)


def layer_forward(P: Dict[str, Tensor], l: int, x: Tensor, vecs: List[Tensor], spec: PathSpec,
                  drop: Dropper, trace: Optional[Dict] = None) -> Tensor:
    """One Meta_Transformer_Layer; models/satrans.py:50-100."""
    pre = f"domain_int_layers.{l}."
    D, H = spec.embedding_dim, spec.head_num
    d = D // H
    psz = spec.meta_param_size
    q = x @ P[pre + "W_Query"]                                        # :55
    k = x @ P[pre + "W_Key"]                                          # :56
    v = x @ P[pre + "W_Value"]                                        # :57
    if "Q" in spec.meta_mode:                                         # :60-66
        if "gate" in spec.flag:
            q = q * vecs[0].unsqueeze(1) * 2
        elif "bilinear" in spec.flag:
            pass
        else:
            q = metanet(q, vecs[0][:, :psz], P[pre + "Q_meta_mlp.ffn_layer_norm.weight"],
                        P[pre + "Q_meta_mlp.ffn_layer_norm.bias"], spec, drop, l, "metaQ")
    if "K" in spec.meta_mode:                                         # :67-73
        if "gate" in spec.flag:
            k = k * vecs[1].unsqueeze(1) * 2
        elif "bilinear" in spec.flag:
            pass
        else:
            k = metanet(k, vecs[1][:, :psz], P[pre + "K_meta_mlp.ffn_layer_norm.weight"],
                        P[pre + "K_meta_mlp.ffn_layer_norm.bias"], spec, drop, l, "metaK")
    B, Fn, _ = x.shape
    # heads are contiguous channel chunks of width d (:75-77) -> [B, H, F, d]
    qh = q.reshape(B, Fn, H, d).permute(0, 2, 1, 3)
    kh = k.reshape(B, Fn, H, d).permute(0, 2, 1, 3)
    vh = v.reshape(B, Fn, H, d).permute(0, 2, 1, 3)
    if "bilinear" in spec.flag:                                       # :79-81
        M = vecs[0].reshape(-1, H, d, d)
        qh = qh @ M
    s = qh @ kh.transpose(-1, -2)                                     # :84
    s = s / (d ** 0.5)                                                # :85-86 (true division)
    a = drop(torch.softmax(s, dim=-1), l, "attn")                     # :87
    o = (a @ vh).permute(0, 2, 1, 3).reshape(B, Fn, D)                # :88-90 heads re-concatenated in order
    o = o @ P[pre + "Out_linear.weight"].t()                          # nn.Linear, no bias
    if "relu" in spec.flag:                                           # :91-92
        o = torch.relu(o)
    o = drop(o, l, "out")                                             # :92,94
    if spec.use_res:
        o = o + x                                                     # :96-97
    y = F.layer_norm(o, (D,), P[pre + "layer_norm.weight"], P[pre + "layer_norm.bias"], 1e-6)  # :99
    if trace is not None:
        trace[f"q{l}"], trace[f"k{l}"], trace[f"v{l}"] = q, k, v
        trace[f"att{l}"] = a.permute(1, 0, 2, 3)                      # reference keeps [H,B,F,F]
        trace[f"out{l}"] = y
    return y


def forward(P: Dict[str, Tensor], X: Tensor, spec: PathSpec, drop: Optional[Dropper] = None,
            trace: Optional[Dict] = None) -> Tuple[Tensor, Tensor]:
    """SATrans.forward; models/satrans.py:197-256.  Returns (probability [B,1], logit [B,1])."""
    drop = drop or Dropper("off")
    x = gather_fields(P, X, spec)
    vecs = scenario_vectors(P, X, spec)
    if trace is not None:
        trace["att_input"] = x
        trace["vec0"] = vecs[0][0]
    for l in range(spec.layer_num):
        x = layer_forward(P, l, x, vecs[l], spec, drop, trace)
    flat = x.flatten(1)                                               # :244
    dense = dense_block(X, spec)
    if dense is not None:
        flat = torch.cat([flat, dense.to(flat.dtype)], dim=-1)        # :247-250
    logit = F.linear(flat, P["dnn_linear.weight"], P["dnn_linear.bias"])  # :254
    return torch.sigmoid(logit), logit                                # :255


# --------------------------------------------------------------------------------------
# Loss, regulariser, optimizer step
# --------------------------------------------------------------------------------------
def regularization_loss(P: Dict[str, Tensor], spec: PathSpec) -> Tensor:
    """sum_w sum(l2 * w^2) over the dnn embedding tables; models/meta_basemodel.py:577-593 with the
    weight groups registered at :179-180 (the linear-model group has l2 = 0 for SATrans,
    models/satrans.py:120, so it contributes nothing)."""
    total = torch.zeros((1,), dtype=next(iter(P.values())).dtype)
    if spec.l2_reg_embedding > 0:
        for name, _ in spec.sparse:
            w = P[f"embedding_dict.{name}.weight"]
            total = total + torch.sum(spec.l2_reg_embedding * torch.square(w))
    return total


def make_leaves(P: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Autograd leaves for a state_dict-shaped dict.  Keys that share storage in the reference
    (K_/V_meta_mlp alias Q_meta_mlp without 'pos', models/satrans.py:46-47; domain_map_dnn_K/V alias
    _Q, :179-180) become ONE leaf so their gradients accumulate as they do there."""
    leaves: Dict[str, Tensor] = {}
    by_ptr: Dict[int, Tensor] = {}
    for k, t in P.items():
        ptr = t.data_ptr()
        if ptr not in by_ptr:
            by_ptr[ptr] = t.detach().clone().requires_grad_(True)
        leaves[k] = by_ptr[ptr]
    return leaves


def loss_and_grads(P: Dict[str, Tensor], X: Tensor, y: Tensor, spec: PathSpec,
                   drop: Optional[Dropper] = None) -> Tuple[float, float, Dict[str, Tensor]]:
    """BCE(sum) + reg, then autograd; models/meta_basemodel.py:314-327.

    Returns (bce_sum, reg, grads by key).  Keys that receive no gradient on this path
    (linear_model.*, out.bias) are absent from the dict.
    """
    leaves = make_leaves(P)
    prob, _ = forward(leaves, X, spec, drop)
    bce = F.binary_cross_entropy(prob.squeeze(-1), y.to(prob.dtype).reshape(-1), reduction="sum")
    reg = regularization_loss(leaves, spec)
    total = bce + reg.sum()
    total.backward()
    grads = {k: t.grad for k, t in leaves.items() if t.grad is not None}
    return float(bce.detach()), float(reg.detach().sum()), grads


class OracleTrainer:
    """Dense torch.optim.Adam over every tensor, as reference main.py:343 does
    (`Adam(model.parameters(), lr)`, dense embedding grads because `sparse=False` at
    models/meta_basemodel.py:168).  Tensors that never receive a gradient are skipped by Adam
    exactly as in the reference (grad is None)."""

    def __init__(self, P: Dict[str, Tensor], spec: PathSpec, lr: float,
                 betas=(0.9, 0.999), eps: float = 1e-8, optimizer: str = "adam", loss: str = "binary_crossentropy"):
        self.spec = spec
        self.leaves = make_leaves(P)
        self.loss = loss
        uniq = list({id(t): t for t in self.leaves.values()}.values())
        # the optimizers models/meta_basemodel.py:612-640 resolves, with torch's own arithmetic
        if optimizer == "adam":
            self.optim = torch.optim.Adam(uniq, lr=lr, betas=betas, eps=eps)
        elif optimizer == "sgd":
            self.optim = torch.optim.SGD(uniq, lr=lr)
        elif optimizer == "adagrad":
            self.optim = torch.optim.Adagrad(uniq, lr=lr)
        elif optimizer == "rmsprop":
            self.optim = torch.optim.RMSprop(uniq, lr=lr)
        else:
            raise ValueError(optimizer)

    def step(self, X: Tensor, y: Tensor, drop: Optional[Dropper] = None, return_prob: bool = False):
        prob, _ = forward(self.leaves, X, self.spec, drop)
        self.optim.zero_grad()
        fn = {"binary_crossentropy": F.binary_cross_entropy, "mse": F.mse_loss, "mae": F.l1_loss}[self.loss]
        bce = fn(prob.squeeze(-1), y.to(prob.dtype).reshape(-1), reduction="sum")       # models/meta_basemodel.py:317,642-653
        reg = regularization_loss(self.leaves, self.spec)
        (bce + reg.sum()).backward()
        self.optim.step()
        if return_prob:
            return prob.detach().squeeze(-1)
        return float(bce.detach()), float(reg.detach().sum())

    def state(self) -> Dict[str, Tensor]:
        return {k: t.detach() for k, t in self.leaves.items()}


# --------------------------------------------------------------------------------------
# Counter-based dropout masks: numpy restatement of satrans_amd/csrc/rng.h so that a test can
# hand the HIP kernels' exact masks to `Dropper('masks')`.
# --------------------------------------------------------------------------------------
def _mix32(x):
    """Same integer hash as `satrans_mix32` in satrans_amd/csrc/rng.h (uint32 arithmetic)."""
    import numpy as np
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def dropout_keep(seed: int, step: int, layer: int, site: int, sample, elem, p: float):
    """Boolean keep-mask for element `elem` of sample `sample` (numpy arrays broadcast together).

    site: 0 = MetaNet-Q output [F*D], 1 = MetaNet-K output [F*D], 2 = attention probabilities
    [H*F*F], 3 = output projection [F*D].  keep <=> 24-bit uniform >= p * 2^24.
    """
    import numpy as np
    with np.errstate(over="ignore"):
        key = _mix32(np.uint32(seed) ^ (np.uint32(step) * np.uint32(0x9E3779B9)))
        key = _mix32(key ^ np.uint32((layer * 4 + site + 1) * 0x85EBCA6B & 0xFFFFFFFF))
        h = _mix32(np.asarray(sample, dtype=np.uint32) * np.uint32(0xC2B2AE35) ^ key)
        # blocks of four consecutive elements (rng.h: drop_block_hash, drop_keep4): one multiply round on key + block * C, the
        # top 24 bits are the word of the block's first element, one step of a full-period 24-bit LCG each of the next three
        elem = np.asarray(elem, dtype=np.uint32)
        t = (h + ((elem >> np.uint32(2)) & np.uint32(0xFFFFFF)) * np.uint32(0x9E3779)).astype(np.uint32)
        t ^= t >> np.uint32(15)
        t = (t * np.uint32(0x2C1B3C6D)).astype(np.uint32)
        t ^= t >> np.uint32(12)
        u, r = np.broadcast_arrays(t >> np.uint32(8), elem & np.uint32(3))
        u = u.astype(np.uint64)
        for k in (1, 2, 3):
            nxt = (u * np.uint64(0xF1EA5D) + np.uint64(0x3C6EF3)) & np.uint64(0xFFFFFF)
            u = np.where(r >= k, nxt, u)
    return u >= np.uint64(int(p * 16777216.0))


def dropout_masks(seed: int, step: int, B: int, Fn: int, D: int, H: int, L: int, p: float) -> Dict:
    """All masks of one forward pass, scaled by 1/(1-p), in the shapes `layer_forward` multiplies."""
    import numpy as np
    masks = {}
    scale = 1.0 / (1.0 - p)
    b = np.arange(B, dtype=np.uint32)
    for l in range(L):
        for site, name in ((0, "metaQ"), (1, "metaK"), (3, "out")):
            e = np.arange(Fn * D, dtype=np.uint32)
            keep = dropout_keep(seed, step, l, site, b[:, None], e[None, :], p)
            masks[(l, name)] = torch.from_numpy((keep * scale).astype(np.float32)).reshape(B, Fn, D)
        # attention rows are padded to a multiple of four keys in the element index (rng.h: drop_attn_elem)
        fp = (Fn + 3) & ~3
        e = (np.arange(H * Fn, dtype=np.uint32)[:, None] * np.uint32(fp) + np.arange(Fn, dtype=np.uint32)[None, :]).reshape(-1)
        keep = dropout_keep(seed, step, l, 2, b[:, None], e[None, :], p)
        masks[(l, "attn")] = torch.from_numpy((keep * scale).astype(np.float32)).reshape(B, H, Fn, Fn)
    return masks


def bce_sum(prob: Tensor, y: Tensor) -> Tensor:
    """F.binary_cross_entropy(reduction='sum') with torch's log clamp at -100."""
    return F.binary_cross_entropy(prob.reshape(-1), y.to(prob.dtype).reshape(-1), reduction="sum")


# --------------------------------------------------------------------------------------
# Sibling users of the same kernels; test infrastructure like everything above
# --------------------------------------------------------------------------------------
def selfattention_layer(P: Dict[str, Tensor], x: Tensor, head_num: int, use_res: bool = True, scaling: bool = True,
                        drop: Optional[Dropper] = None) -> Tuple[Tensor, Tensor]:
    """models/submodules.py:209-236 (SelfAttention_Layer.forward): q,k,v = x W; heads = contiguous chunks of D/H channels;
    softmax(q k^T / sqrt(d)) -> dropout -> @ v; heads concatenated in order; dropout; += x W_Res; ReLU; LayerNorm(eps 1e-6).
    -> (y [B,F,D], normalized_att_scores [H,B,F,F])"""
    drop = drop or Dropper("off")
    D = x.shape[-1]
    d = D // head_num
    q, k, v = x @ P["W_Query"], x @ P["W_Key"], x @ P["W_Value"]
    split = lambda t: torch.stack(torch.split(t, d, dim=2))
    qh, kh, vh = split(q), split(k), split(v)
    s = torch.einsum('bnik,bnjk->bnij', qh, kh)
    if scaling:
        s = s / d ** 0.5
    att = drop(torch.softmax(s, dim=-1), 0, "attn")
    o = torch.cat(torch.split(torch.matmul(att, vh), 1), dim=-1).squeeze(0)
    o = drop(o, 0, "out")
    if use_res:
        o = o + x @ P["W_Res"]
    o = torch.relu(o)
    return F.layer_norm(o, (D,), P["layer_norm.weight"], P["layer_norm.bias"], 1e-6), att


def meta_transformation(P: Dict[str, Tensor], ids: Tensor, x: Tensor, units: List[int], use_norm: bool,
                        drop: Optional[Dropper] = None) -> Tensor:
    """models/basemodel.py:191-199 + MetaNet (models/submodules.py:77-103): per-sample generated weights
    vec = Linear(relu(domain_embeddings[id])); y = [LN](dropout(relu(x @ W1) @ W2) + x)."""
    drop = drop or Dropper("off")
    vec = F.linear(torch.relu(P["domain_embeddings.weight"][ids.long()]), P["domain_map_dnn.weight"], P["domain_map_dnn.bias"])
    D, U = units[0], units[1]
    W1 = vec[:, :D * U].reshape(-1, D, U)
    W2 = vec[:, D * U:D * U + U * units[2]].reshape(-1, U, units[2])
    y = drop(torch.relu(x @ W1) @ W2, 0, "meta_q") + x
    if use_norm:
        y = F.layer_norm(y, (D,), P["ffn_layer_norm.weight"], P["ffn_layer_norm.bias"], 1e-6)
    return y
