"""Structural helpers of deepctr_torch.inputs, restated here on their own (documented behaviour of deepctr-torch
0.2.9, SURVEY.md §8c "Shim surface").  Nothing is imported from the product package, so a column-layout bug in
satrans_amd/inputs.py cannot leak into the golden X that the reference is fed while the fixtures are recorded."""
from collections import OrderedDict, namedtuple

import torch.nn as nn


class SparseFeat(namedtuple("SparseFeat", ["name", "vocabulary_size", "embedding_dim", "use_hash", "dtype",
                                           "embedding_name", "group_name"])):
    __slots__ = ()

    def __new__(cls, name, vocabulary_size, embedding_dim=4, use_hash=False, dtype="int32", embedding_name=None,
                group_name="default_group"):
        if embedding_name is None:
            embedding_name = name
        if embedding_dim == "auto":
            embedding_dim = 6 * int(pow(vocabulary_size, 0.25))
        return super().__new__(cls, name, vocabulary_size, embedding_dim, use_hash, dtype, embedding_name, group_name)

    def __hash__(self):
        return self.name.__hash__()


class DenseFeat(namedtuple("DenseFeat", ["name", "dimension", "dtype"])):
    __slots__ = ()

    def __new__(cls, name, dimension=1, dtype="float32"):
        return super().__new__(cls, name, dimension, dtype)

    def __hash__(self):
        return self.name.__hash__()


class VarLenSparseFeat(namedtuple("VarLenSparseFeat", ["sparsefeat", "maxlen", "combiner", "length_name"])):
    __slots__ = ()

    def __new__(cls, sparsefeat, maxlen, combiner="mean", length_name=None):
        return super().__new__(cls, sparsefeat, maxlen, combiner, length_name)

    name = property(lambda self: self.sparsefeat.name)
    vocabulary_size = property(lambda self: self.sparsefeat.vocabulary_size)
    embedding_dim = property(lambda self: self.sparsefeat.embedding_dim)
    embedding_name = property(lambda self: self.sparsefeat.embedding_name)

    def __hash__(self):
        return self.name.__hash__()


def build_input_features(feature_columns):
    """name -> (start, end) column span of X: one column per sparse feature, `dimension` per dense feature, `maxlen`
    (+1 with a length column) per variable-length feature; a name already placed is skipped."""
    spans, start = OrderedDict(), 0
    for col in feature_columns:
        if col.name in spans:
            continue
        if isinstance(col, SparseFeat):
            spans[col.name] = (start, start + 1)
            start += 1
        elif isinstance(col, DenseFeat):
            spans[col.name] = (start, start + col.dimension)
            start += col.dimension
        elif isinstance(col, VarLenSparseFeat):
            spans[col.name] = (start, start + col.maxlen)
            start += col.maxlen
            if col.length_name is not None and col.length_name not in spans:
                spans[col.length_name] = (start, start + 1)
                start += 1
        else:
            raise TypeError("Invalid feature column type,got", type(col))
    return spans


def get_feature_names(feature_columns):
    return list(build_input_features(feature_columns).keys())


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
