"""Data-parallel exchange for the training step: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm, over xGMI inside a node).

The reference's only parallelism is an optional single-process `torch.nn.DataParallel` wrap
(models/meta_basemodel.py:272-275) whose semantics are: `batch_size` is PER GPU, parameters are replicated, the
loss is a SUM over samples, gradients are summed onto one device.  The same semantics here, one exchange step
per iteration:

  * dense parameters (~0.15-0.28 M floats, latency-bound): ONE all_reduce(SUM) of the flat gradient buffer;
  * embedding gradients: all_gather of the (arena row id int32, gradient row D x fp32) pairs each rank produced
    (B*F pairs, ~10.6 MB at B=4096).  Every rank then sorts and applies the SAME merged list in the SAME order
    (rank-major, then position), so the replicas stay bit-identical without ever moving a table-sized buffer.
    The regulariser gradient 2*l2*p is identical on every rank and is added locally, after the exchange.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _host_staged(t: torch.Tensor) -> bool:
    """gloo moves host memory: device tensors are staged through the host (CPU tests, and the two-ranks-on-one-GPU
    parity test; production runs use nccl = RCCL, which takes device pointers)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_gather(out: torch.Tensor, inp: torch.Tensor) -> None:
    if _host_staged(inp):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.cpu())
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, inp)


def _all_reduce(t: torch.Tensor, op=dist.ReduceOp.SUM) -> None:
    if _host_staged(t):
        host = t.cpu()
        dist.all_reduce(host, op=op)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op)


def exchange(flat_grad: torch.Tensor, rows: torch.Tensor, gemb: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-reduce `flat_grad` in place (SUM); return (all rows [W*n], all gradient rows [W*n, D]) in rank order."""
    w = world_size()
    if w == 1:
        return rows.reshape(-1), gemb.reshape(-1, gemb.shape[-1])
    _all_reduce(flat_grad)
    rows = rows.reshape(-1).contiguous()
    gemb = gemb.reshape(rows.numel(), -1).contiguous()
    all_rows = torch.empty(w * rows.numel(), dtype=rows.dtype, device=rows.device)
    all_gemb = torch.empty(w * gemb.shape[0], gemb.shape[1], dtype=gemb.dtype, device=gemb.device)
    _all_gather(all_rows, rows)
    _all_gather(all_gemb, gemb)
    return all_rows, all_gemb


def gather_rows(rows: torch.Tensor) -> torch.Tensor:
    """All-gather of the arena row ids every rank gathered this step ([W*n] int32, rank-major).  Issued right after
    the gather kernel so that the sort (and the streaming Adam that only needs the touched-row bitmap) can start
    while the layers are still computing."""
    w = world_size()
    rows = rows.reshape(-1).contiguous()
    if w == 1:
        return rows
    out = torch.empty(w * rows.numel(), dtype=rows.dtype, device=rows.device)
    _all_gather(out, rows)
    return out


def all_reduce_flat(flat_grad: torch.Tensor) -> None:
    """SUM over ranks, in place, of the flat gradient buffer (dense parameters + the dense gradient of the small
    embedding tables, which the engine keeps at its tail): the loss is a sum over samples, so gradients add."""
    if world_size() > 1:
        _all_reduce(flat_grad)


def gather_grad_rows(gemb: torch.Tensor) -> torch.Tensor:
    """All-gather of gradient rows ([W*n, D], rank-major: the same order as `gather_rows`)."""
    w = world_size()
    gemb = gemb.reshape(-1, gemb.shape[-1]).contiguous()
    if w == 1:
        return gemb
    out = torch.empty(w * gemb.shape[0], gemb.shape[1], dtype=gemb.dtype, device=gemb.device)
    _all_gather(out, gemb)
    return out


def gather_grad_rows_async(gemb: torch.Tensor):
    """`gather_grad_rows` that returns at once: (out, handle).  The caller keeps issuing independent work on its stream
    and calls `handle.wait()` (None: nothing to wait for) before it reads `out`."""
    w = world_size()
    gemb = gemb.reshape(-1, gemb.shape[-1]).contiguous()
    if w == 1:
        return gemb, None
    out = torch.empty(w * gemb.shape[0], gemb.shape[1], dtype=gemb.dtype, device=gemb.device)
    if _host_staged(gemb):
        _all_gather(out, gemb)
        return out, None
    return out, dist.all_gather_into_tensor(out, gemb, async_op=True)


def exchange_grads(flat_grad: torch.Tensor, gemb: torch.Tensor) -> torch.Tensor:
    """All-reduce `flat_grad` in place (SUM) and all-gather the gradient rows ([W*n, D], rank-major: the same order
    as `gather_rows`)."""
    w = world_size()
    gemb = gemb.reshape(-1, gemb.shape[-1]).contiguous()
    if w == 1:
        return gemb
    _all_reduce(flat_grad)
    out = torch.empty(w * gemb.shape[0], gemb.shape[1], dtype=gemb.dtype, device=gemb.device)
    _all_gather(out, gemb)
    return out


def all_reduce_scalars(t: torch.Tensor) -> torch.Tensor:
    """Sum of per-rank scalars (losses, counts) for logging."""
    if world_size() > 1:
        _all_reduce(t)
    return t
