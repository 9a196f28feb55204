"""SATrans with the reference's constructor / forward signatures, evaluated by HIP kernels.

Reference: models/satrans.py:105-256 (model), :13-48 (per-layer parameters).  Parameters are created by the
same sequence of torch calls as the reference, so for a given seed `state_dict()` is
bit-identical to the reference's and checkpoints written by the reference (`main.py:399-401`) load unchanged.
The arithmetic of `forward` is not in this file: see satrans_amd/engine.py and satrans_amd/csrc/.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from .basemodel import BaseModel, make_embedding_tables
from .inputs import DenseFeat, SparseFeat, split_columns


class _MetaNetNorm(nn.Module):
    """Parameter holder for reference `MetaNet` (models/submodules.py:64-75): only its LayerNorm has parameters."""

    def __init__(self, hidden_dim):
        super().__init__()
        self.ffn_layer_norm = nn.LayerNorm(hidden_dim, eps=1e-6)


class MetaTransformerLayerParams(nn.Module):
    """Parameters of one Meta_Transformer_Layer, created in the reference's order (models/satrans.py:32-47):
    W_Query/W_Key/W_Value [D,D] (y = x @ W), Out_linear (nn.Linear without bias), layer_norm (eps 1e-6); then
    EVERY parameter so far - LayerNorm gamma and beta included - is drawn from N(0, 0.05); the MetaNet norms are
    created afterwards and keep gamma = 1, beta = 0.  Without 'pos' in the flag K_/V_meta_mlp alias Q_meta_mlp."""

    def __init__(self, embedding_size, flag, head_num):
        super().__init__()
        if head_num <= 0:
            raise ValueError('head_num must be a int > 0')
        if embedding_size % head_num != 0:
            raise ValueError('embedding_size is not an integer multiple of head_num!')
        self.W_Query = nn.Parameter(torch.empty(embedding_size, embedding_size))
        self.W_Key = nn.Parameter(torch.empty(embedding_size, embedding_size))
        self.W_Value = nn.Parameter(torch.empty(embedding_size, embedding_size))
        self.Out_linear = nn.Linear(embedding_size, embedding_size, bias=False)
        self.layer_norm = nn.LayerNorm(embedding_size, eps=1e-6)
        for tensor in self.parameters():
            nn.init.normal_(tensor, mean=0.0, std=0.05)
        self.Q_meta_mlp = _MetaNetNorm(embedding_size)
        self.K_meta_mlp = self.Q_meta_mlp if 'pos' not in flag else _MetaNetNorm(embedding_size)
        self.V_meta_mlp = self.Q_meta_mlp if 'pos' not in flag else _MetaNetNorm(embedding_size)
        self.normalized_att_scores = None        # filled by forward when attention capture is on


class _ScenarioEncoder(nn.Module):
    """`DNN_v2(in, [P])` = exactly one Linear(in -> P), weight N(0, 1e-4), default-uniform bias
    (models/submodules.py:31-43)."""

    def __init__(self, in_dim, out_dim, init_std=0.0001):
        super().__init__()
        self.linears = nn.ModuleList([nn.Linear(in_dim, out_dim)])
        nn.init.normal_(self.linears[0].weight, mean=0, std=init_std)


class SATrans(BaseModel):
    def __init__(self, linear_feature_columns, dnn_feature_columns, domain_column_list,
                 num_domains_list, att_layer_num=2, domain_att_layer_num=1,
                 att_head_num=2,
                 share_domain_dnn_across_layers=False,
                 use_domain_dnn_linear=False,
                 att_res=True,
                 use_linear=True,
                 use_dnn=False,
                 dnn_hidden_units=(256, 128), dnn_activation='relu',
                 meta_dnn_hidden_units=(64, 32),
                 meta_mode='Q',
                 l2_reg_dnn=0, l2_reg_embedding=1e-5, dnn_use_bn=False, dnn_dropout=0, init_std=0.0001, seed=1024,
                 task='binary', device='cpu', gpus=None, flag=None):
        # The reference calls BaseModel.__init__ WITHOUT the flag (models/satrans.py:120-122), so the base class's
        # 'noembinit' test (models/meta_basemodel.py:114) never fires for SATrans: the tables are always re-drawn
        # from N(0, init_std).  The flag is stored afterwards for the layer variants.
        super().__init__(linear_feature_columns, dnn_feature_columns, l2_reg_linear=0,
                         l2_reg_embedding=l2_reg_embedding, init_std=init_std, seed=seed, task=task,
                         device=device, gpus=gpus, flag=None)
        if not isinstance(flag, str):
            raise TypeError("`flag` must be a string (the reference tests substrings of it, e.g. 'pos' in flag)")
        self.flag = flag
        sparse, dense, varlen = split_columns(dnn_feature_columns)
        if varlen:
            raise NotImplementedError("VarLenSparseFeat columns (reference main.py never passes any)")
        self.use_linear, self.use_dnn = use_linear, use_dnn
        self.num_domains_list = num_domains_list
        self.use_domain_dnn_linear = use_domain_dnn_linear
        self.share_domain_dnn_across_layers = share_domain_dnn_across_layers
        self.dnn_hidden_units = dnn_hidden_units
        self.domain_att_layer_num = domain_att_layer_num
        self.att_layer_num = att_layer_num
        self.att_head_num = att_head_num
        self.att_res = att_res
        self.meta_mode = meta_mode
        self.domain_column_list = domain_column_list
        embedding_size = self.embedding_size
        field_num = len(self.embedding_dict)
        if field_num != len(sparse):
            # the reference sizes dnn_linear by the number of distinct tables and then fails with a shape error in
            # forward when two columns share an embedding_name; fail at construction instead of reading past the weight
            raise ValueError(f"{len(sparse)} sparse columns share {field_num} embedding tables: SATrans needs one "
                             f"embedding_name per column (dnn_linear is sized by the number of tables)")
        dense_in = sum(c.dimension for c in linear_feature_columns if isinstance(c, DenseFeat))
        self.domain_embedding_dim = embedding_size

        self.domain_embeddings = nn.Embedding(num_domains_list[0] + 1, self.domain_embedding_dim)
        units = [embedding_size] + [int(u) for u in meta_dnn_hidden_units]
        self.meta_dnn_hidden_units = units
        self.domain_int_layers = nn.ModuleList(
            [MetaTransformerLayerParams(embedding_size, flag, att_head_num) for _ in range(domain_att_layer_num)])
        meta_param_size = sum(units[i] * units[i + 1] for i in range(len(units) - 1))
        if 'bilinear' in flag:
            meta_param_size = (embedding_size ** 2) // att_head_num
        elif 'gate' in flag:
            meta_param_size = embedding_size
        self.meta_param_size = meta_param_size

        if 'pos' in flag:
            self.domain_embedding_dim *= 2
            self.layerid_embeddings = nn.Embedding(domain_att_layer_num, self.domain_embedding_dim // 2)
            self.qkvid_embeddings = nn.Embedding(3, self.domain_embedding_dim // 2)
        if 'onlyemb' in flag:
            self.domain_embeddings = nn.Embedding(num_domains_list[0] + 1, meta_param_size)
        else:
            self.domain_map_dnn_Q = _ScenarioEncoder(self.domain_embedding_dim, meta_param_size)
            self.domain_map_dnn_K = self.domain_map_dnn_Q
            self.domain_map_dnn_V = self.domain_map_dnn_Q
        self.dnn_linear = nn.Linear(field_num * embedding_size + dense_in, 1)
        if len(domain_column_list) > 1:
            self.domain_feature_columns = [c for c in linear_feature_columns if c.name in domain_column_list]
            self.domain_embedding_dict = make_embedding_tables(dnn_feature_columns, init_std)

        self.capture_attention = False           # set True to fill layer.normalized_att_scores ([H,B,F,F])
        self.to(device)
        self._rebind_storage()

    # the tensors that receive gradients on this path, in a fixed order (linear_model.*, out.bias
    # and, under 'pos', V_meta_mlp.* never do, so Adam never touches them - as in the reference where their
    # grad stays None)
    def _trainable_flat(self):
        flat = OrderedDict()
        flag = self.flag
        # scenario embedding: one table, or (several scenario columns, satrans.py:205-207) the mean of those columns'
        # rows in the second table set `domain_embedding_dict`; `domain_embeddings` then receives no gradient
        multi = len(self.domain_column_list) > 1
        scen_keys = [f"domain_embedding_dict.{c.embedding_name}.weight" for c in self.domain_feature_columns] \
            if multi else ["domain_embeddings.weight"]
        for key in scen_keys:
            flat[key] = self.get_parameter(key)
        for l, layer in enumerate(self.domain_int_layers):
            pre = f"domain_int_layers.{l}."
            flat[pre + "W_Query"] = layer.W_Query
            flat[pre + "W_Key"] = layer.W_Key
            flat[pre + "W_Value"] = layer.W_Value
            flat[pre + "Out_linear.weight"] = layer.Out_linear.weight
            flat[pre + "layer_norm.weight"] = layer.layer_norm.weight
            flat[pre + "layer_norm.bias"] = layer.layer_norm.bias
            uses_metanet = 'gate' not in flag and 'bilinear' not in flag
            if uses_metanet and 'Q' in self.meta_mode:
                flat[pre + "Q_meta_mlp.ffn_layer_norm.weight"] = layer.Q_meta_mlp.ffn_layer_norm.weight
                flat[pre + "Q_meta_mlp.ffn_layer_norm.bias"] = layer.Q_meta_mlp.ffn_layer_norm.bias
            if uses_metanet and 'K' in self.meta_mode and layer.K_meta_mlp is not layer.Q_meta_mlp:
                flat[pre + "K_meta_mlp.ffn_layer_norm.weight"] = layer.K_meta_mlp.ffn_layer_norm.weight
                flat[pre + "K_meta_mlp.ffn_layer_norm.bias"] = layer.K_meta_mlp.ffn_layer_norm.bias
            if uses_metanet and 'K' in self.meta_mode and 'Q' not in self.meta_mode \
                    and layer.K_meta_mlp is layer.Q_meta_mlp:
                flat[pre + "Q_meta_mlp.ffn_layer_norm.weight"] = layer.Q_meta_mlp.ffn_layer_norm.weight
                flat[pre + "Q_meta_mlp.ffn_layer_norm.bias"] = layer.Q_meta_mlp.ffn_layer_norm.bias
        modulated = ('Q' in self.meta_mode) or ('K' in self.meta_mode) or ('bilinear' in flag)
        if 'pos' in flag and modulated:
            flat["layerid_embeddings.weight"] = self.layerid_embeddings.weight
            flat["qkvid_embeddings.weight"] = self.qkvid_embeddings.weight
        if 'onlyemb' not in flag and modulated:
            flat["domain_map_dnn_Q.linears.0.weight"] = self.domain_map_dnn_Q.linears[0].weight
            flat["domain_map_dnn_Q.linears.0.bias"] = self.domain_map_dnn_Q.linears[0].bias
        if not modulated:
            for key in scen_keys:
                del flat[key]
        flat["dnn_linear.weight"] = self.dnn_linear.weight
        flat["dnn_linear.bias"] = self.dnn_linear.bias
        return flat

    def _require_engine(self):
        if self._engine is None:
            from .engine import PathEngine
            self._engine = PathEngine(self)
            pend = getattr(self, "_pending_opt_state", None)
            if pend is not None:                      # optimizer state carried over a .to() or loaded for a resume
                self._engine.load_optimizer_state(pend)
                self._pending_opt_state = None
        return self._engine

    def set_forward_precision(self, precision: str) -> None:
        """"fp32" (default, the parity path) or "bf16": evaluation forwards (predict / evaluate / eval-mode calls) then run
        their dense products on the bf16 matrix pipe with fp32 accumulation (BASELINE.json configs[1]); training is always fp32."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self._require_engine().fwd_bf16 = precision == "bf16"

    def forward(self, X):
        """X: FloatTensor [B, C] in `feature_index` column order -> probabilities [B, 1]
        (reference models/satrans.py:197-256).  Dropout is applied when the module is in training mode."""
        engine = self._require_engine()
        y = engine.forward(X, training=self.training, capture_attention=self.capture_attention)
        engine.raise_if_bad_ids()
        return y
