"""Feature-column types and the X column layout.

The reference takes these from the third-party package deepctr-torch 0.2.9
(`from deepctr_torch.inputs import SparseFeat, DenseFeat, VarLenSparseFeat,
get_feature_names` at reference main.py:5; `build_input_features` at
models/meta_basemodel.py:26,164).  deepctr-torch is not installed on the GPU
box, so the drop-in ships its own with the same constructor signatures and the
same column-layout rule:

  * one X column per SparseFeat, in declaration order,
  * `dimension` columns per DenseFeat,
  * `maxlen` columns per VarLenSparseFeat (+1 when `length_name` is set),
  * a name that was already placed is skipped (main.py passes
    linear_feature_columns + dnn_feature_columns, i.e. every column twice).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Iterable, List

DEFAULT_GROUP_NAME = "default_group"


class SparseFeat:
    """A categorical field backed by one embedding table of `vocabulary_size` rows."""

    __slots__ = ("name", "vocabulary_size", "embedding_dim", "use_hash", "dtype",
                 "embedding_name", "group_name")

    def __init__(self, name, vocabulary_size, embedding_dim=4, use_hash=False, dtype="int32",
                 embedding_name=None, group_name=DEFAULT_GROUP_NAME):
        if embedding_dim == "auto":
            embedding_dim = 6 * int(pow(vocabulary_size, 0.25))
        if use_hash:
            raise NotImplementedError("feature hashing is not supported (the reference never enables it)")
        self.name = name
        self.vocabulary_size = int(vocabulary_size)
        self.embedding_dim = int(embedding_dim)
        self.use_hash = use_hash
        self.dtype = dtype
        self.embedding_name = name if embedding_name is None else embedding_name
        self.group_name = group_name

    def _key(self):
        return ("sparse", self.name)

    def __hash__(self):
        return hash(self.name)

    def __eq__(self, other):
        return isinstance(other, SparseFeat) and self._key() == other._key()

    def __repr__(self):
        return (f"SparseFeat(name={self.name!r}, vocabulary_size={self.vocabulary_size}, "
                f"embedding_dim={self.embedding_dim}, embedding_name={self.embedding_name!r})")


class DenseFeat:
    """A real-valued field occupying `dimension` float columns of X."""

    __slots__ = ("name", "dimension", "dtype")

    def __init__(self, name, dimension=1, dtype="float32"):
        self.name = name
        self.dimension = int(dimension)
        self.dtype = dtype

    def __hash__(self):
        return hash(self.name)

    def __eq__(self, other):
        return isinstance(other, DenseFeat) and self.name == other.name

    def __repr__(self):
        return f"DenseFeat(name={self.name!r}, dimension={self.dimension})"


class VarLenSparseFeat:
    """A padded id list pooled into one embedding (accepted for API parity;
    reference main.py:103,146 always passes an empty list of these)."""

    __slots__ = ("sparsefeat", "maxlen", "combiner", "length_name")

    def __init__(self, sparsefeat, maxlen, combiner="mean", length_name=None):
        self.sparsefeat = sparsefeat
        self.maxlen = int(maxlen)
        self.combiner = combiner
        self.length_name = length_name

    name = property(lambda self: self.sparsefeat.name)
    vocabulary_size = property(lambda self: self.sparsefeat.vocabulary_size)
    embedding_dim = property(lambda self: self.sparsefeat.embedding_dim)
    use_hash = property(lambda self: self.sparsefeat.use_hash)
    dtype = property(lambda self: self.sparsefeat.dtype)
    embedding_name = property(lambda self: self.sparsefeat.embedding_name)
    group_name = property(lambda self: self.sparsefeat.group_name)

    def __hash__(self):
        return hash(self.name)

    def __eq__(self, other):
        return isinstance(other, VarLenSparseFeat) and self.name == other.name

    def __repr__(self):
        return f"VarLenSparseFeat({self.sparsefeat!r}, maxlen={self.maxlen}, combiner={self.combiner!r})"


def build_input_features(feature_columns: Iterable) -> "OrderedDict[str, tuple]":
    """name -> (start, end) column span of X, in first-seen order."""
    spans: "OrderedDict[str, tuple]" = OrderedDict()
    cursor = 0
    for col in feature_columns:
        if col.name in spans:
            continue
        if isinstance(col, SparseFeat):
            width = 1
        elif isinstance(col, DenseFeat):
            width = col.dimension
        elif isinstance(col, VarLenSparseFeat):
            width = col.maxlen
        else:
            raise TypeError(f"Invalid feature column type, got {type(col)}")
        spans[col.name] = (cursor, cursor + width)
        cursor += width
        if isinstance(col, VarLenSparseFeat) and col.length_name is not None \
                and col.length_name not in spans:
            spans[col.length_name] = (cursor, cursor + 1)
            cursor += 1
    return spans


def get_feature_names(feature_columns: Iterable) -> List[str]:
    return list(build_input_features(feature_columns).keys())


def split_columns(feature_columns):
    """(sparse, dense, varlen) lists, each in declaration order."""
    cols = list(feature_columns) if feature_columns else []
    sparse = [c for c in cols if isinstance(c, SparseFeat)]
    dense = [c for c in cols if isinstance(c, DenseFeat)]
    varlen = [c for c in cols if isinstance(c, VarLenSparseFeat)]
    return sparse, dense, varlen


class PackedInput:
    """Ids and dense features of a batch as TWO device matrices: `ids` [N, C] integer (every feature_index column, the dense
    ones unused) and `dense` [N, n_dense] float32.  Used when a vocabulary does not survive the reference's fp32 id matrix
    (2**24 rows and more, models/meta_basemodel.py:311) AND the model has DenseFeat columns; slices like a tensor."""

    def __init__(self, ids, dense):
        self.ids, self.dense = ids, dense
        self.shape = ids.shape

    def __getitem__(self, s):
        return PackedInput(self.ids[s], self.dense[s])

    def index_select(self, dim, idx):
        return PackedInput(self.ids.index_select(dim, idx), self.dense.index_select(dim, idx))

    def __len__(self):
        return self.ids.shape[0]
