"""The per-step training metrics of `fit(verbose > 0)` evaluated where the predictions are, without a device sync.

The reference computes sklearn metrics on host copies of every batch (models/meta_basemodel.py:330-337:
`metric_fun(y.cpu().numpy(), y_pred.cpu().numpy().astype("float64"))`), i.e. one device-to-host round trip plus a few
milliseconds of sklearn per step - more than the whole training step takes here.  These functions return 0-dim float64
tensors on the inputs' device with the same values (tests/test_host_cpu.py compares them with sklearn, ties included);
`fit` keeps them on the device and reads them once per epoch.  They are metrics, not part of the hot path: plain
torch ops.
"""
from __future__ import annotations

import torch


def log_loss(y: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """sklearn.metrics.log_loss(y, p.astype(float64)): probabilities clipped to [eps, 1 - eps] with float64 eps, mean of
    -(y log p + (1 - y) log(1 - p))."""
    y64, p64 = y.double().reshape(-1), p.double().reshape(-1)
    eps = torch.finfo(torch.float64).eps
    p64 = p64.clamp(eps, 1.0 - eps)
    return -(y64 * torch.log(p64) + (1.0 - y64) * torch.log(1.0 - p64)).mean()


def roc_auc(y: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """sklearn.metrics.roc_auc_score(y, p): the Mann-Whitney statistic with tied scores counted one half (the area under
    the trapezoidal ROC curve over the distinct thresholds).  NaN when only one class is present (sklearn raises there)."""
    y64, s = y.double().reshape(-1), p.reshape(-1).contiguous()
    s_sorted, order = torch.sort(s)
    y_sorted = y64[order]
    # average 1-based rank of every group of tied scores, without data-dependent shapes (nothing here makes the host wait):
    # a score's group occupies the sorted positions [left, right)
    left = torch.searchsorted(s_sorted, s_sorted, right=False)
    right = torch.searchsorted(s_sorted, s_sorted, right=True)
    ranks = (left + right + 1).double() * 0.5
    n_pos = y_sorted.sum()
    n_neg = y_sorted.numel() - n_pos
    u = (ranks * y_sorted).sum() - n_pos * (n_pos + 1.0) / 2.0
    return u / (n_pos * n_neg)


def mse(y: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    d = y.double().reshape(-1) - p.double().reshape(-1)
    return (d * d).mean()


def accuracy(y: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    return ((p.reshape(-1) > 0.5).double() == y.double().reshape(-1)).double().mean()


def per_domain_auc(y: torch.Tensor, p: torch.Tensor, domain_ids: torch.Tensor):
    """The evaluation report of reference main.py:355-374 on the device: overall ROC AUC, then one AUC per scenario id from
    the smallest to the largest id present (`for i in range(ids.min(), ids.max() + 1)`), plus the test loss
    F.binary_cross_entropy(pred, labels.double()) of main.py:359 (mean; float64, log clamped at -100 as torch does).
    -> (auc, {id: auc}, loss) as Python floats (one host read at the end of an evaluation; not a per-step metric)."""
    y, p, ids = y.reshape(-1), p.reshape(-1), domain_ids.reshape(-1).long()
    overall = roc_auc(y, p)
    lo, hi = int(ids.min()), int(ids.max())
    per = {}
    for i in range(lo, hi + 1):
        sel = ids == i
        per[i] = roc_auc(y[sel], p[sel])
    p64, y64 = p.double(), y.double()
    loss = -(y64 * torch.log(p64).clamp_min(-100.0) + (1.0 - y64) * torch.log(1.0 - p64).clamp_min(-100.0)).mean()
    vals = torch.stack([overall, loss] + [per[i] for i in range(lo, hi + 1)]).cpu().tolist()
    return vals[0], {i: vals[2 + k] for k, i in enumerate(range(lo, hi + 1))}, vals[1]


FUSED = ("binary_crossentropy", "logloss", "auc")       # what satrans_batch_metrics computes


def fused_supported(names, n: int, y: torch.Tensor, p: torch.Tensor) -> bool:
    """One launch of csrc/metrics.hip serves this batch: log loss and / or ROC AUC of at most 8,192 fp32 rows on the GPU."""
    return (bool(names) and all(k in FUSED for k in names) and 0 < n <= 8192 and y.is_cuda and p.is_cuda
            and y.dtype == torch.float32 and p.dtype == torch.float32)


def fused_logloss_auc(y: torch.Tensor, p: torch.Tensor, out: torch.Tensor) -> None:
    """out[0] = log_loss(y, p), out[1] = roc_auc(y, p) (same values as the functions above, sklearn's) by ONE kernel launch
    (csrc/metrics.hip: the dozen torch launches of the two functions cost the host more than a training step takes).
    `out`: two float64 on the device; nothing is read back."""
    from . import native as N
    y, p = y.reshape(-1).contiguous(), p.reshape(-1).contiguous()
    N.check(N.lib().satrans_batch_metrics(y.data_ptr(), p.data_ptr(), y.numel(), out.data_ptr(), N.stream_handle(y.device)),
            "satrans_batch_metrics")


BY_NAME = {"binary_crossentropy": log_loss, "logloss": log_loss, "auc": roc_auc, "mse": mse, "accuracy": accuracy,
           "acc": accuracy}
