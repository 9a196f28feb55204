"""Structural helpers of deepctr_torch.inputs, backed by this repo's own column types."""
import torch.nn as nn
from satrans_amd.inputs import (SparseFeat, DenseFeat, VarLenSparseFeat, build_input_features,  # noqa: F401
                                get_feature_names)


def create_embedding_matrix(feature_columns, init_std=0.0001, linear=False, sparse=False, device="cpu"):
    cols = [c for c in feature_columns if isinstance(c, (SparseFeat, VarLenSparseFeat))]
    cols = [c for c in cols if isinstance(c, SparseFeat)] + [c for c in cols if isinstance(c, VarLenSparseFeat)]
    table = nn.ModuleDict({c.embedding_name: nn.Embedding(c.vocabulary_size, 1 if linear else c.embedding_dim,
                                                          sparse=sparse) for c in cols})
    for t in table.values():
        nn.init.normal_(t.weight, mean=0, std=init_std)
    return table.to(device)


def varlen_embedding_lookup(X, embedding_dict, sequence_input_dict, varlen_sparse_feature_columns):
    if varlen_sparse_feature_columns:
        raise NotImplementedError("shim: variable-length features are not on the SATrans path")
    return {}


def get_varlen_pooling_list(embedding_dict, features, feature_index, varlen_sparse_feature_columns, device):
    if varlen_sparse_feature_columns:
        raise NotImplementedError("shim: variable-length features are not on the SATrans path")
    return []


def combined_dnn_input(sparse_embedding_list, dense_value_list):
    raise NotImplementedError("shim: not on the SATrans path")
