"""The few names of deepctr_torch.layers the SATrans path touches."""
import torch
import torch.nn as nn


def concat_fun(inputs, axis=-1):
    return inputs[0] if len(inputs) == 1 else torch.cat(inputs, dim=axis)


def activation_layer(act_name, hidden_size=None, dice_dim=2):
    if isinstance(act_name, str) and act_name.lower() == "relu":
        return nn.ReLU(inplace=True)
    raise NotImplementedError(f"shim: activation {act_name!r}")


class PredictionLayer(nn.Module):
    def __init__(self, task="binary", use_bias=True, **kwargs):
        super().__init__()
        if task not in ("binary", "multiclass", "regression"):
            raise ValueError("task must be binary,multiclass or regression")
        self.task, self.use_bias = task, use_bias
        if use_bias:
            self.bias = nn.Parameter(torch.zeros((1,)))

    def forward(self, X):
        out = X + self.bias if self.use_bias else X
        return torch.sigmoid(out) if self.task == "binary" else out


class _OffPath(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("shim: this deepctr layer is not on the SATrans path")


DNN = InteractingLayer = _OffPath
