"""World-size-2 test of the data-parallel exchange (satrans_amd/parallel.py) on the gloo backend.

Each rank takes half of a golden batch, produces its local dense gradients and (arena row, gradient row) pairs with
the CPU oracle, runs the exchange, and checks what the training step relies on:
  * the all-reduced flat gradient equals the full-batch gradient (the loss is a SUM over samples),
  * the gathered (row, gradient) lists are the rank-major concatenation, identical on both ranks bit for bit,
  * scattering the merged list reproduces the full-batch dense table gradient.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import satrans_oracle as O
from tests.helpers import Case


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_grads(case, X, y):
    """(flat gradient dict of the non-table tensors, arena rows [n], gradient rows [n, D]) of one shard."""
    spec = case.spec()
    P = case.tensors("param")
    leaves = O.make_leaves(P)
    x = O.gather_fields(leaves, X, spec).detach().requires_grad_(True)
    vecs = O.scenario_vectors(leaves, X, spec)
    h = x
    for l in range(spec.layer_num):
        h = O.layer_forward(leaves, l, h, vecs[l], spec, O.Dropper("off"))
    logit = torch.nn.functional.linear(h.flatten(1), leaves["dnn_linear.weight"], leaves["dnn_linear.bias"])
    O.bce_sum(torch.sigmoid(logit), y).backward()
    flat = {k: t.grad.clone() for k, t in leaves.items() if t.grad is not None and not k.startswith("embedding_dict.")}
    offs = np.concatenate([[0], np.cumsum(case.meta["vocab"])])
    rows = torch.stack([X[:, col].long() + int(offs[i]) for i, (_, col) in enumerate(spec.sparse)], dim=1)
    return flat, rows.reshape(-1).to(torch.int32), x.grad.reshape(-1, x.shape[-1]).contiguous()


def _worker(rank, world, port, name, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from satrans_amd import parallel
        torch.set_num_threads(1)
        case = Case(name)
        B = case.X.shape[0] // world
        Xs, ys = case.X[rank * B:(rank + 1) * B], case.y[rank * B:(rank + 1) * B]
        flat, rows, gemb = _local_grads(case, Xs, ys)
        keys = sorted(flat)
        flat_buf = torch.cat([flat[k].reshape(-1) for k in keys])
        # the three calls of engine.train_step's exchange, in its order
        all_rows = parallel.gather_rows(rows)
        parallel.all_reduce_flat(flat_buf)
        all_gemb, pending = parallel.gather_grad_rows_async(gemb)
        if pending is not None:
            pending.wait()
        assert parallel.world_size() == world and parallel.rank() == rank and parallel.exchange_enabled()
        assert parallel.STATS["all_gather_rows_i32"]["bytes_out"] == world * rows.numel() * 4
        assert parallel.STATS["all_gather_grad_rows_f32"]["bytes_out"] == world * gemb.numel() * 4
        assert parallel.STATS["all_reduce_flat_f32"]["calls"] == 1
        torch.save(dict(keys=keys, flat=flat_buf, rows=all_rows, gemb=all_gemb, local_rows=rows, local_gemb=gemb),
                   os.path.join(out, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["small_qkv"])
def test_exchange_world_size_2(tmp_path, name):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rank{k}.pt")) for k in range(world)]
    # replicas see the same merged data, bit for bit, in rank-major order
    for key in ("flat", "rows", "gemb"):
        assert torch.equal(r[0][key], r[1][key]), key
    assert torch.equal(r[0]["rows"], torch.cat([r[0]["local_rows"], r[1]["local_rows"]]))
    assert torch.equal(r[0]["gemb"], torch.cat([r[0]["local_gemb"], r[1]["local_gemb"]]))
    # ... and that data is the full-batch gradient
    case = Case(name)
    n = (case.X.shape[0] // world) * world
    flat, rows, gemb = _local_grads(case, case.X[:n], case.y[:n])
    want = torch.cat([flat[k].reshape(-1) for k in r[0]["keys"]])
    np.testing.assert_allclose(r[0]["flat"].numpy(), want.numpy(), rtol=0, atol=2e-6 * float(want.abs().max()))
    R, D = sum(case.meta["vocab"]), case.meta["D"]
    dense_merged = torch.zeros(R, D).index_add_(0, r[0]["rows"].long(), r[0]["gemb"])
    dense_full = torch.zeros(R, D).index_add_(0, rows.long(), gemb)
    np.testing.assert_allclose(dense_merged.numpy(), dense_full.numpy(), rtol=0, atol=2e-6 * float(dense_full.abs().max()))


def test_single_process_is_a_no_op():
    from satrans_amd import parallel
    assert parallel.world_size() == 1 and parallel.rank() == 0
    rows = torch.arange(6, dtype=torch.int32).reshape(2, 3)
    g = torch.randn(6, 4)
    flat = torch.randn(5)
    before = flat.clone()
    assert not parallel.exchange_enabled()
    assert torch.equal(parallel.gather_rows(rows), rows.reshape(-1))
    parallel.all_reduce_flat(flat)
    out, pending = parallel.gather_grad_rows_async(g)
    assert torch.equal(out, g) and pending is None and torch.equal(flat, before)
