"""Fixture loading shared by the CPU and GPU tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

from oracle.satrans_oracle import PathSpec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and not f.startswith("sibling_"))
SIBLING_SELFATT = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("sibling_selfatt"))
SIBLING_METANET = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("sibling_metanet"))
TRAIN_CASES = [c for c in ALL_CASES if c != "small_relu"]


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.z = z
        self.meta = json.loads(str(z["meta"]))
        self.X = torch.from_numpy(z["X"])
        self.y = torch.from_numpy(z["y"])

    def tensors(self, prefix, dtype=torch.float32):
        """state_dict-shaped dict; aliased keys share ONE tensor object, as in the reference."""
        out = {k[len(prefix) + 1:]: torch.from_numpy(self.z[k]).to(dtype) for k in self.z.files
               if k.startswith(prefix + "/")}
        for k in self.z.files:
            if k.startswith("alias/"):
                out[k[6:]] = out[str(self.z[k])]
        return out

    def arrays(self, prefix):
        return {k[len(prefix) + 1:]: self.z[k] for k in self.z.files if k.startswith(prefix + "/")}

    def spec(self) -> PathSpec:
        m = self.meta
        col = {n: i for i, n in enumerate(m["feature_names"])}
        return PathSpec(
            sparse=[(f, col[f]) for f in m["fields"]],
            dense=[(col[f], col[f] + 1) for f in m["dense"]],
            domain_cols=[col[c] for c in m["domain"]],
            embedding_dim=m["D"], head_num=m["H"], layer_num=m["L"], flag=m["flag"], meta_mode=m["mode"],
            meta_units=[m["D"]] + list(m["units"]),
            multi_domain_sparse=[(f, col[f]) for f in m["fields"] if f in m["domain"]],
        )

    def columns(self):
        """Feature columns of the product package for this case (reference main.py:182-191)."""
        from satrans_amd.inputs import SparseFeat, DenseFeat
        m = self.meta
        return [SparseFeat(f, vocabulary_size=v, embedding_dim=m["D"]) for f, v in zip(m["fields"], m["vocab"])] + \
               [DenseFeat(f, 1) for f in m["dense"]]


def build_model(case: "Case", device: str):
    """The product model for a golden case, constructed the way reference main.py:292-306 does."""
    from satrans_amd import SATrans
    m = case.meta
    cols = case.columns()
    model = SATrans(linear_feature_columns=cols, dnn_feature_columns=cols, domain_column_list=list(m["domain"]),
                    num_domains_list=m["num_domains_list"], att_layer_num=0, domain_att_layer_num=m["L"],
                    att_head_num=m["H"], share_domain_dnn_across_layers=False, use_domain_dnn_linear=False,
                    use_linear=False, meta_mode=m["mode"], use_dnn=False, meta_dnn_hidden_units=tuple(m["units"]),
                    seed=m["seed"], device=device, flag=m["flag"])
    return model


def _adam_steps(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return json.loads(str(z["meta"]))["adam_steps"]


ADAM_CASES = [c for c in TRAIN_CASES if _adam_steps(c) > 0]     # cases that carry parameters / moments after Adam steps
NATIVE_CASES = list(ALL_CASES)
NATIVE_TRAIN_CASES = [c for c in NATIVE_CASES if c in TRAIN_CASES]
NATIVE_ADAM_CASES = [c for c in NATIVE_CASES if c in ADAM_CASES]
