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


def _owner_worker(rank, world, port, out):
    """The collectives of the owner-form step (engine._train_step_owner) with synthetic rows: ids to their owners, the owners'
    values back, gradient rows to the owners, then the slice broadcasts that bring the replicas together."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from satrans_amd import parallel
        torch.set_num_threads(1)
        R, D, n = 1000, 4, 257                                        # table rows, row width, positions per rank
        g = torch.Generator().manual_seed(100 + rank)
        # skewed on purpose: rank r asks mostly for rows of owner (r + 1) % world, and nobody asks owner 3 for anything but rank 2
        hot = (rank + 1) % world
        rows = torch.cat([torch.randint(hot * 250, hot * 250 + 250, (n - 40,), generator=g),
                          torch.randint(0, 750, (40,), generator=g)]).to(torch.int32)
        rows, _ = torch.sort(rows)
        table = torch.arange(R * D, dtype=torch.float32).reshape(R, D)          # the same replica everywhere
        bounds = torch.tensor([250, 500, 750], dtype=torch.int32)
        cut = torch.searchsorted(rows, bounds)
        edges = torch.cat([cut.new_zeros(1), cut, cut.new_full((1,), n)])
        counts = parallel.gather_counts(edges[1:] - edges[:-1])
        send, recv = counts[rank].tolist(), counts[:, rank].tolist()
        assert sum(send) == n and counts.shape == (world, world)
        got_ids = parallel.all_to_all_rows(rows, send, recv, "ids")
        assert got_ids.numel() == sum(recv)
        assert bool(((got_ids >= rank * 250) & (got_ids < rank * 250 + 250)).all()), "an owner received a row outside its slice"
        vals = table[got_ids.long()] + 0.5 * rank                     # the owner's (current) values: tagged with the owner
        back = parallel.all_to_all_rows(vals, recv, send, "values")
        want = table[rows.long()] + 0.5 * (rows // 250).float()[:, None]
        assert torch.equal(back, want), "values did not come back in the order the ids were sent"
        grads = torch.randn(n, D, generator=g)
        recv_g = parallel.all_to_all_rows(grads, send, recv, "grads")
        # owner-side dense sum of what it received, then replica sync by slice broadcasts
        dense = torch.zeros(R, D).index_add_(0, got_ids.long(), recv_g)
        for o in range(world):
            parallel.broadcast_slice(dense[o * 250:(o + 1) * 250], o)
        torch.save(dict(rows=rows, grads=grads, dense=dense, stats={k: dict(v) for k, v in parallel.STATS.items()}),
                   os.path.join(out, f"owner{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_owner_exchange_world_size_4(tmp_path):
    """World size 4 on gloo: uneven all-to-alls sized from the exchanged counts (one owner receives nothing from three of the
    ranks), values returned in request order, and slice broadcasts after which every replica holds the sum over ALL ranks."""
    world = 4
    mp.spawn(_owner_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"owner{k}.pt")) for k in range(world)]
    full = torch.zeros(1000, 4)
    for k in range(world):                                             # rank-major accumulation = the owners' receive order
        full.index_add_(0, r[k]["rows"].long(), r[k]["grads"])
    for k in range(world):
        assert torch.equal(r[k]["dense"], r[0]["dense"]), "replicas differ after the slice broadcasts"
    np.testing.assert_allclose(r[0]["dense"].numpy(), full.numpy(), rtol=0, atol=1e-5)
    assert r[0]["stats"]["ids"]["calls"] == 1 and r[0]["stats"]["broadcast_table_slices_f32"]["calls"] == world
