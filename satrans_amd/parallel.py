"""Data-parallel exchange for the training step: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm, over xGMI inside a node).

The reference's only parallelism is an optional single-process `torch.nn.DataParallel` wrap
(models/meta_basemodel.py:272-275) whose semantics are: `batch_size` is PER GPU, parameters are replicated, the
loss is a SUM over samples, gradients are summed onto one device.  The same semantics here, one exchange step
per iteration:

  * dense parameters and the SMALL embedding tables (replicated on every rank): ONE all_reduce(SUM) of the flat gradient buffer
    (2.8 MB at AliCCP size), the same dense step on every rank;
  * LARGE embedding tables, row-ownership form (default): every rank owns a slice of the rows - values, Adam moments, the lazy
    optimizer's bookkeeping - and three uneven all-to-alls per step move row ids to their owners (a step ahead, on a process group
    of its own: `prefetch_group`), current rows back, gradient rows to the owners (`all_to_all_rows`; engine._train_step_owner);
  * LARGE tables, replicated form (SATRANS_DP_MODE=replicated, round 2): all_gather of the (arena row id int32, gradient row
    D x fp32) pairs each rank produced; every rank sorts and applies the SAME merged list in the SAME order (rank-major, then
    position), so the replicas stay bit-identical without ever moving a table-sized buffer.
The regulariser gradient 2*l2*p needs no exchange: it is a function of the row itself and is added where the row is stepped.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def exchange_enabled() -> bool:
    """True when the training step runs its exchange branch: more than one rank, or SATRANS_FORCE_EXCHANGE=1 with an
    initialised process group of ONE rank (every collective is then an identity, but it really goes through the backend
    - this is how a one-GPU box exercises RCCL: device-pointer int32 all-gather, async fp32 all-gather + wait, SUM
    all-reduce; tests/test_gpu_parity.py::test_rccl_single_rank_exchange...)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("SATRANS_FORCE_EXCHANGE", "0") == "1"


# bytes this rank handed to / received from each collective since the last reset (bench.py reports them per step)
STATS = {}


def _count(name: str, sent: int, received: int) -> None:
    st = STATS.setdefault(name, {"calls": 0, "bytes_in": 0, "bytes_out": 0})
    st["calls"] += 1
    st["bytes_in"] += int(sent)
    st["bytes_out"] += int(received)


def reset_stats() -> None:
    STATS.clear()


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


def gather_rows(rows: torch.Tensor) -> torch.Tensor:
    """All-gather of the arena row ids every rank gathered this step ([W*n] int32, rank-major).  Issued right after
    the gather kernel so that the sort (and the streaming Adam that only needs the touched-row bitmap) can start
    while the layers are still computing."""
    w = world_size()
    rows = rows.reshape(-1).contiguous()
    if not exchange_enabled():
        return rows
    out = torch.empty(w * rows.numel(), dtype=rows.dtype, device=rows.device)
    _all_gather(out, rows)
    _count("all_gather_rows_i32", rows.numel() * rows.element_size(), out.numel() * out.element_size())
    return out


def all_reduce_flat(flat_grad: torch.Tensor) -> None:
    """SUM over ranks, in place, of the flat gradient buffer (dense parameters + the dense gradient of the small
    embedding tables, which the engine keeps at its tail): the loss is a sum over samples, so gradients add."""
    if exchange_enabled():
        _all_reduce(flat_grad)
        _count("all_reduce_flat_f32", flat_grad.numel() * 4, flat_grad.numel() * 4)


def gather_grad_rows_async(gemb: torch.Tensor):
    """`gather_grad_rows` that returns at once: (out, handle).  The caller keeps issuing independent work on its stream
    and calls `handle.wait()` (None: nothing to wait for) before it reads `out`."""
    w = world_size()
    gemb = gemb.reshape(-1, gemb.shape[-1]).contiguous()
    if not exchange_enabled():
        return gemb, None
    out = torch.empty(w * gemb.shape[0], gemb.shape[1], dtype=gemb.dtype, device=gemb.device)
    _count("all_gather_grad_rows_f32", gemb.numel() * 4, out.numel() * 4)
    if _host_staged(gemb):
        _all_gather(out, gemb)
        return out, None
    return out, dist.all_gather_into_tensor(out, gemb, async_op=True)


# ---- row-ownership exchange (engine: SATRANS_DP_MODE=owner) -----------------------------------------------------------------
def gather_counts(counts: torch.Tensor) -> torch.Tensor:
    """[W] int64 per-destination counts of this rank -> [W, W] on the HOST (row r = what rank r sends to each owner).  The one
    device-to-host read of the owner-mode step: all_to_all_single needs its split sizes as Python ints."""
    w = world_size()
    counts = counts.to(torch.int64).reshape(-1)
    if not exchange_enabled():
        return counts.reshape(1, -1).cpu()
    out = torch.empty(w * counts.numel(), dtype=torch.int64, device=counts.device)
    _all_gather(out, counts)
    _count("all_gather_counts_i64", counts.numel() * 8, out.numel() * 8)
    return out.reshape(w, -1).cpu()


_PREFETCH_GROUP = None


def prefetch_group():
    """A second process group (its own RCCL communicator and stream) for the exchange a step issues AHEAD - the next batch's row
    ids to their owners, on the engine's side stream under the current step's tail: on the default group it would queue behind
    the current step's gradient collectives, whose inputs are ready later.  Created on first use: COLLECTIVE (every rank takes
    its first owner-form step at the same point of the program)."""
    global _PREFETCH_GROUP
    if _PREFETCH_GROUP is None and dist.is_available() and dist.is_initialized():
        _PREFETCH_GROUP = dist.new_group()
        if dist.get_backend() == "nccl" and torch.cuda.is_available():
            # RCCL creates a communicator at the group's FIRST collective (a blocking rendezvous of all ranks): do that here, with
            # nothing else in flight, rather than in the middle of a step beside the default group's collectives
            warm = torch.zeros(1, device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(warm, group=_PREFETCH_GROUP)
            torch.cuda.synchronize()
    return _PREFETCH_GROUP


def all_to_all_rows(inp: torch.Tensor, send_splits, recv_splits, name: str, out: torch.Tensor = None, group=None) -> torch.Tensor:
    """Uneven all-to-all along dim 0: rows [sum(send_splits), ...] -> [sum(recv_splits), ...]; chunk o of `inp` goes to rank o,
    chunk r of the result came from rank r.  RCCL: grouped point-to-point over xGMI (every pair has its own link).
    `out`: a caller-owned buffer of at least sum(recv_splits) rows (its head is filled and returned); `group`: the process group
    (default: the world)."""
    send_splits, recv_splits = [int(v) for v in send_splits], [int(v) for v in recv_splits]
    inp = inp.contiguous()
    if out is None:
        out = torch.empty((sum(recv_splits),) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
    else:
        out = out[:sum(recv_splits)]
    row_bytes = inp.element_size()
    for dim in inp.shape[1:]:
        row_bytes *= int(dim)
    _count(name, sum(send_splits) * row_bytes, sum(recv_splits) * row_bytes)
    if not exchange_enabled() or (world_size() == 1 and dist.get_backend() == "gloo"):
        out.copy_(inp)
        return out
    if _host_staged(inp):
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, inp.cpu(), recv_splits, send_splits, group=group)
        out.copy_(h_out)
    else:
        dist.all_to_all_single(out, inp, recv_splits, send_splits, group=group)
    return out


def broadcast_slice(t: torch.Tensor, src: int) -> None:
    """In-place broadcast of one owner's slice of a replicated table (replica synchronisation at flush points)."""
    if not exchange_enabled() or world_size() == 1:
        return
    _count("broadcast_table_slices_f32", t.numel() * t.element_size() if rank() == src else 0,
           t.numel() * t.element_size() if rank() != src else 0)
    if _host_staged(t):
        host = t.cpu()
        dist.broadcast(host, src)
        if rank() != src:
            t.copy_(host)
    else:
        dist.broadcast(t, src)


def all_reduce_scalars(t: torch.Tensor) -> torch.Tensor:
    """Sum of per-rank scalars (losses, counts) for logging."""
    if exchange_enabled():
        _all_reduce(t)
    return t
