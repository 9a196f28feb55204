"""Helper process of tests/test_gpu_parity.py::test_rccl_single_rank_exchange_is_bitwise_the_local_step.

    python tests/rccl_one_rank.py nccl|plain <case> <steps> <out.pt> <port> [small_table_rows]

`nccl`: initialises a ONE-rank process group on the nccl backend (= RCCL) BEFORE anything touches the GPU, sets
SATRANS_FORCE_EXCHANGE=1 so that the training step runs its multi-rank branch, and trains `steps` steps (dropout on).
`plain`: the same steps with the same table classes (SATRANS_SPLIT_TABLES=1) and no process group.
Both write every state_dict tensor, the Adam moments and the collective byte counters."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, name, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    os.environ["SATRANS_SPLIT_TABLES"] = "1"
    # 20: some golden tables small, some large; 0: every table "large", so the rank-major list of the exchange has exactly
    # B * F entries - the length of this rank's own [B, F] row matrix (it must NOT take the per-field sort)
    os.environ["SATRANS_SMALL_TABLE_ROWS"] = sys.argv[6] if len(sys.argv) > 6 else "20"
    import torch
    import torch.distributed as dist
    if mode == "nccl":
        os.environ["SATRANS_FORCE_EXCHANGE"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", sys.argv[5])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from satrans_amd import parallel
    from tests.helpers import Case, build_model
    c = Case(name)
    model = build_model(c, "cuda:0")
    model.compile(torch.optim.Adam(model.parameters(), lr=c.meta["lr"]), "binary_crossentropy")
    model.train()
    eng = model._require_engine()
    X, y = c.X.to("cuda:0"), c.y.to("cuda:0")
    for _ in range(steps):
        eng.train_step(X, y)
    torch.cuda.synchronize()
    res = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    opt = model.optimizer_state_dict()
    for k, st in opt["state"].items():
        res["exp_avg/" + k], res["exp_avg_sq/" + k] = st["exp_avg"], st["exp_avg_sq"]
    res["__stats__"] = {k: dict(v) for k, v in parallel.STATS.items()}
    res["__exchange__"] = parallel.exchange_enabled()
    res["__backend__"] = dist.get_backend() if dist.is_initialized() else None
    torch.save(res, out)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
