"""Sibling users of the attention-path kernels, for the reference's baselines that plug them in.

  * `SelfAttention_Layer` - reference models/submodules.py:178-238 (`usetrans` in models/star.py:70-72, mmoe.py:77-79,
    ple.py:72-74, sharedbottom.py:65-67, adasparse.py:145-147): same constructor, same parameter creation order and init
    (W_Query, W_Key, W_Value, W_Out, layer_norm, W_Res ~ N(0, 0.05); W_Out is never used by the reference's forward either).
  * `MetaTransformation` - reference BaseModel.meta_transformation (models/basemodel.py:191-199) with its MetaNet
    (models/submodules.py:64-103): the scenario embedding, the one-Linear scenario encoder and the MetaNet over the embedding
    block.  The generated weights are tabulated per SCENARIO ([S,P], S rows) instead of per sample ([B,P]).

Both are ordinary `nn.Module`s whose forward/backward are HIP launches (csrc/layer_generic.hip) wrapped in a
`torch.autograd.Function`, so they can sit inside any torch model.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import native as N

NO_SCALING, NO_NORM = 128, 256


class _DropClock:
    """Counter-based dropout needs a (seed, step) pair per forward; the step advances with every training forward."""

    def __init__(self):
        self.seed = int(torch.initial_seed() & 0xFFFFFFFF)
        self.step = 0


class _SelfAttFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wq, wk, wv, wres, ln_w, ln_b, mod):
        lib = N.lib()
        N.require_gpu(x, "SelfAttention_Layer")
        x = x.contiguous().float()
        B, F, D = x.shape
        d = N.SelfAttDesc()
        d.B, d.F, d.D, d.H = B, F, D, mod.head_num
        d.flags = (N.TRAIN if mod.training else 0) | (0 if mod.use_res else N.NO_RES) | (0 if mod.scaling else NO_SCALING)
        d.layer, d.drop_p = 0, 0.1
        if mod.training:
            mod._clock.step += 1
        d.seed, d.step = mod._clock.seed, mod._clock.step & 0xFFFFFFFF
        ln = torch.cat([ln_w.detach().reshape(-1), ln_b.detach().reshape(-1)]).contiguous()
        d.x, d.w_query, d.w_key, d.w_value = x.data_ptr(), wq.data_ptr(), wk.data_ptr(), wv.data_ptr()
        d.w_res = wres.data_ptr() if wres is not None else None
        d.ln_g, d.ln_b = ln.data_ptr(), ln.data_ptr() + 4 * D
        n = int(lib.satrans_selfatt_saved_floats(C.byref(d)))
        if n < 0:
            raise N.NativeError(f"SelfAttention_Layer: shape B={B} F={F} D={D} H={mod.head_num} is not supported")
        saved = torch.empty(n, dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        att = None
        if mod.capture_attention:
            att = torch.empty(mod.head_num, B, F, F, dtype=torch.float32, device=x.device)
        N.check(lib.satrans_selfatt_fwd(C.byref(d), y.data_ptr(), att.data_ptr() if att is not None else None, saved.data_ptr(),
                                        N.stream_handle(x.device)), "satrans_selfatt_fwd")
        mod.normalized_att_scores = att
        ctx.desc, ctx.mod = d, mod
        ctx.save_for_backward(x, wq, wk, wv, wres if wres is not None else x.new_empty(0), ln, saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = N.lib()
        x, wq, wk, wv, wres, ln, saved = ctx.saved_tensors
        d = ctx.desc
        D = d.D
        use_res = wres.numel() > 0
        scratch = torch.empty(int(lib.satrans_selfatt_scratch_floats(C.byref(d))), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        g = [torch.zeros(D, D, dtype=torch.float32, device=x.device) for _ in range(4)]
        g_ln = torch.zeros(2, D, dtype=torch.float32, device=x.device)
        N.check(lib.satrans_selfatt_bwd(C.byref(d), dy.contiguous().data_ptr(), dx.data_ptr(), saved.data_ptr(), scratch.data_ptr(),
                                        g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr() if use_res else None,
                                        g_ln.data_ptr(), N.stream_handle(x.device)), "satrans_selfatt_bwd")
        return dx, g[0], g[1], g[2], (g[3] if use_res else None), g_ln[0], g_ln[1], None


class SelfAttention_Layer(nn.Module):
    def __init__(self, embedding_size, head_num=2, use_res=True, scaling=True, seed=1024, device='cpu'):
        super().__init__()
        if head_num <= 0:
            raise ValueError('head_num must be a int > 0')
        if embedding_size % head_num != 0:
            raise ValueError('embedding_size is not an integer multiple of head_num!')
        self.att_embedding_size = embedding_size // head_num
        self.head_num, self.use_res, self.scaling, self.seed = head_num, use_res, scaling, seed
        self.W_Query = nn.Parameter(torch.empty(embedding_size, embedding_size))
        self.W_Key = nn.Parameter(torch.empty(embedding_size, embedding_size))
        self.W_Value = nn.Parameter(torch.empty(embedding_size, embedding_size))
        self.W_Out = nn.Parameter(torch.empty(embedding_size, embedding_size))
        self.layer_norm = nn.LayerNorm(embedding_size, eps=1e-6)
        if self.use_res:
            self.W_Res = nn.Parameter(torch.empty(embedding_size, embedding_size))
        for tensor in self.parameters():
            nn.init.normal_(tensor, mean=0.0, std=0.05)
        self.normalized_att_scores = None
        self.capture_attention = False
        self._clock = _DropClock()
        self.to(device)

    def forward(self, inputs):
        if len(inputs.shape) != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (len(inputs.shape)))
        return _SelfAttFn.apply(inputs, self.W_Query, self.W_Key, self.W_Value, self.W_Res if self.use_res else None,
                                self.layer_norm.weight, self.layer_norm.bias, self)


class _MetaNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, ln_w, ln_b, scenario_ids, mod):
        lib = N.lib()
        N.require_gpu(x, "MetaTransformation")
        x = x.contiguous().float()
        table = table.contiguous().float()
        B, F, D = x.shape
        S = table.shape[0]
        dev = x.device
        i32 = dict(dtype=torch.int32, device=dev)
        sid_in = scenario_ids.to(torch.int32).contiguous()
        sid, order, seg = torch.empty(B, **i32), torch.empty(B, **i32), torch.empty(S + 1, **i32)
        status = torch.zeros(1, **i32)
        bucket = torch.empty(int(lib.satrans_bucket_workspace_bytes(B, S)), dtype=torch.uint8, device=dev)
        st = N.stream_handle(dev)
        N.check(lib.satrans_bucket_scenarios(sid_in.data_ptr(), N.ID_I32, 1, 0, B, S, sid.data_ptr(), order.data_ptr(), seg.data_ptr(),
                                             status.data_ptr(), bucket.data_ptr(), bucket.numel(), st), "satrans_bucket_scenarios")
        d = N.MetaNetDesc()
        d.B, d.F, d.D, d.U, d.S = B, F, D, mod.units[1], S
        d.flags = (N.TRAIN if mod.training else 0) | (0 if mod.use_norm else NO_NORM)
        d.layer, d.drop_p = 0, 0.1
        if mod.training:
            mod._clock.step += 1
        d.seed, d.step = mod._clock.seed, mod._clock.step & 0xFFFFFFFF
        d.tab_stride = table.shape[1]
        ln = torch.cat([ln_w.detach().reshape(-1), ln_b.detach().reshape(-1)]).contiguous() if mod.use_norm else None
        d.x, d.order, d.seg, d.tab = x.data_ptr(), order.data_ptr(), seg.data_ptr(), table.data_ptr()
        d.ln_g = ln.data_ptr() if ln is not None else None
        d.ln_b = ln.data_ptr() + 4 * D if ln is not None else None
        n = int(lib.satrans_metanet_saved_floats(C.byref(d)))
        if n < 0:
            raise N.NativeError(f"MetaTransformation: shape D={D} U={mod.units[1]} is not supported")
        saved = torch.empty(n, dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        N.check(lib.satrans_metanet_fwd(C.byref(d), y.data_ptr(), saved.data_ptr(), st), "satrans_metanet_fwd")
        if int(status.item()) != 0:
            raise IndexError("index out of range in self: a scenario id exceeds the scenario table")
        ctx.desc, ctx.mod = d, mod
        ctx.save_for_backward(x, table, ln if ln is not None else x.new_empty(0), saved, order, seg)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = N.lib()
        x, table, ln, saved, order, seg = ctx.saved_tensors
        d = ctx.desc
        scratch = torch.empty(int(lib.satrans_metanet_scratch_floats(C.byref(d))), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        g_tab = torch.zeros_like(table)
        g_ln = torch.zeros(2, d.D, dtype=torch.float32, device=x.device)
        N.check(lib.satrans_metanet_bwd(C.byref(d), dy.contiguous().data_ptr(), dx.data_ptr(), saved.data_ptr(), scratch.data_ptr(),
                                        g_tab.data_ptr(), g_ln.data_ptr() if ln.numel() else None, N.stream_handle(x.device)),
                "satrans_metanet_bwd")
        return dx, g_tab, (g_ln[0] if ln.numel() else None), (g_ln[1] if ln.numel() else None), None, None


class MetaTransformation(nn.Module):
    """`BaseModel.meta_transformation` as a module: scenario ids [B] + embedding block [B,F,D] -> MetaNet(block, weights of the
    sample's scenario).  Parameters in the reference's creation order (models/basemodel.py:137-149): domain_embeddings
    [num_domains+1, D] (torch default N(0,1)), domain_map_dnn = one Linear(D -> P) with weight N(0, 1e-4), meta_net LayerNorm
    (use_norm: the reference's flag 'metanorm')."""

    def __init__(self, embedding_dim, num_domains, meta_dnn_hidden_units=(32, 64, 32), use_norm=False, init_std=0.0001):
        super().__init__()
        units = [int(u) for u in meta_dnn_hidden_units]
        if len(units) != 3 or units[0] != embedding_dim or units[2] != embedding_dim:
            raise NotImplementedError("meta_dnn_hidden_units must be (D, U, D)")
        self.units, self.use_norm = units, use_norm
        self.domain_embeddings = nn.Embedding(num_domains + 1, embedding_dim)
        p = units[0] * units[1] + units[1] * units[2]
        self.domain_map_dnn = nn.Linear(embedding_dim, p)
        nn.init.normal_(self.domain_map_dnn.weight, mean=0, std=init_std)
        self.ffn_layer_norm = nn.LayerNorm(embedding_dim, eps=1e-6) if use_norm else None
        self._clock = _DropClock()

    def forward(self, scenario_ids, fm_input):
        # [S, P] table: S rows through relu + one Linear (a handful of tiny torch ops with autograd; the per-sample work is HIP)
        table = self.domain_map_dnn(torch.relu(self.domain_embeddings.weight))
        w = self.ffn_layer_norm.weight if self.use_norm else None
        b = self.ffn_layer_norm.bias if self.use_norm else None
        return _MetaNetFn.apply(fm_input, table, w, b, scenario_ids.reshape(-1), self)
