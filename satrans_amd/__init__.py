"""satrans_amd: the SATrans scenario-adaptive attention path as MI355X (gfx950) HIP kernels behind the
reference's Python API.  See DESIGN.md and INTEGRATION.md."""
from .inputs import DenseFeat, SparseFeat, VarLenSparseFeat, build_input_features, get_feature_names  # noqa: F401
from .callbacks import History  # noqa: F401
from .basemodel import BaseModel  # noqa: F401
from .satrans import SATrans  # noqa: F401
from .layers import MetaTransformation, SelfAttention_Layer  # noqa: F401

__all__ = ["SATrans", "BaseModel", "SparseFeat", "DenseFeat", "VarLenSparseFeat", "get_feature_names",
           "build_input_features", "History", "SelfAttention_Layer", "MetaTransformation"]
