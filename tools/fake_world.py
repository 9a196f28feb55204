"""Diagnostic: compute-side cost of one rank's step in an N-rank job, measured on ONE GPU.

    python tools/fake_world.py [--mode owner|replicated] [N ...]

`--mode owner` (default, the engine's default for several ranks): rank 0 of N.  The stand-ins deliver what the owner of slice 0
would receive - the rows of N-1 other synthetic batches that fall into its slice, and as many gradient rows - and hand back rows
for the ids this rank asked the other owners for (read from the local arena: the cost of writing the received buffer).
`--mode replicated`: round 2's exchange (below).

`parallel.world_size / gather_rows / gather_grad_rows_async / all_reduce_flat` are replaced by stand-ins that append the arena rows of N-1
other synthetic batches (what the all-gather of ids would deliver) and tile the local gradient rows N times (what
the all-gather of gradient rows would deliver; the tiling copy costs about what writing the received buffer does).
Everything downstream - sort, lazy replay, segmented sums, touched-row Adam over N*B*F rows - is the real code.
No communication time is included: add the all-gather estimate of DESIGN.md to read a scaling number from this.

    python tools/fake_world.py [N ...]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from satrans_amd import parallel  # noqa: E402


def run_owner(N, steps=40, warmup=6, B=8192):
    os.environ["SATRANS_DP_MODE"] = "owner"
    model = bench.build_model("cpu", 0.005)
    model.to("cuda:0"); model.device = "cuda:0"
    eng = model._require_engine()
    model.train()
    n_steps = steps + warmup
    X, y = bench.synth_batches(n_steps * B, 100)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    parallel.world_size = lambda: N
    parallel.rank = lambda: 0
    parallel.exchange_enabled = lambda: True
    parallel.prefetch_group = lambda: None
    eng._owner_world = N                       # (no process group here: skip the first-step flush + group creation)
    bounds = eng._owner_ranges(N, B)
    lo, hi = bounds[0], bounds[1]
    others = [[] for _ in range(n_steps)]
    if N > 1:                                   # rows of the other ranks' batches that fall into slice 0, per step, rank-major
        Xo, _ = bench.synth_batches(n_steps * B * (N - 1), 777)
        Xo = torch.from_numpy(Xo).cuda().long()
        big = eng.row_span[:, 0] >= eng.small_rows
        off = eng.row_span[big, 0][None, :]
        rows_o = (Xo[:, eng.cols.long()[big]] + off).to(torch.int32).reshape(n_steps, N - 1, -1)
        for i in range(n_steps):
            for r in range(N - 1):
                v = rows_o[i, r]
                others[i].append(torch.sort(v[(v >= lo) & (v < hi)])[0])
    # (one tensor per step: the stand-in then costs the host ONE copy, as the real exchange costs it one call)
    others_cat = [torch.cat(o) if o else torch.empty(0, dtype=torch.int32, device="cuda") for o in others]
    arena = model.embedding_arena
    calls = {"ids": 0, "last_ids": None}

    # the epoch plan (what `fit` builds with ONE all-gather per epoch): this rank's split sizes from its own batches, the other
    # ranks' from the stand-in lists - with a plan the id exchange of step t + 1 is prefetched under step t
    def fake_all_gather(out, inp):
        out.zero_()
        out[:inp.numel()] = inp
        o = out.reshape(N, n_steps, N)
        for i in range(n_steps):
            for r in range(1, N):
                o[r, i, 0] = others[i][r - 1].numel()
    parallel._all_gather = fake_all_gather
    eng.plan_owner_counts(Xd, None, B)

    iota = torch.arange(4 * B * 19, dtype=torch.int32, device="cuda")

    def put(dst, src):
        # a gather KERNEL, as the real exchange's receive side is a kernel - `copy_` would go through the runtime's copy path
        # (blit / SDMA), whose interplay with the step's streams is not what this tool measures
        if src.shape[0]:
            torch.index_select(src, 0, iota[:src.shape[0]], out=dst)

    def all_to_all_rows(inp, send, recv, name, out=None, group=None):
        n_recv = int(sum(recv))
        dst = out[:n_recv] if out is not None else torch.empty((n_recv,) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
        # (no allocation in the stand-ins: the caching allocator's cross-stream bookkeeping is not what is measured)
        if name == "all_to_all_row_ids_i32":            # my slice-0 segment + what the other ranks ask owner 0 for
            k = calls["ids"]
            calls["ids"] += 1
            at = int(send[0])
            put(dst[:at], inp[:at])
            put(dst[at:at + others_cat[k].numel()], others_cat[k])
        elif name == "all_to_all_rows_f32":             # values for every id I asked for (the other owners' answers: an arena read)
            torch.index_select(arena, 0, calls["cur_ids"](), out=dst)
        else:                                           # gradient rows: mine for slice 0 + as many rows as the others send
            at = 0
            while at < n_recv:
                take = min(inp.shape[0], n_recv - at)
                put(dst[at:at + take], inp[:take])
                at += take
        return dst

    parallel.all_to_all_rows = all_to_all_rows
    parallel.all_reduce_flat = lambda flat: None
    parallel.broadcast_slice = lambda t, src: None
    parallel.all_reduce_scalars = lambda t: t
    eng.timers = None

    def step(i):
        n_s = B * eng.F_small
        j = (i + 1) % n_steps
        ws = eng.train_workspace(B, 1, False)
        # (the ids this rank asks for: the large-table part of its sorted rows - read when the values come back)
        calls["cur_ids"] = lambda: ws["sorted_rows"][n_s:]
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], next_X=Xd[j * B:(j + 1) * B] if i + 1 < n_steps else None)

    # three passes over the same batches (the plan is positional: re-planned per pass); the FASTEST pass is reported with the
    # host's enqueue time next to it: on a busy host the stand-ins' Python can make the host the bottleneck (enqueue ~ wall), and
    # then the pass says nothing about the GPU
    def prepare_pass():
        eng._prep = None
        calls["ids"] = 0
        eng.plan_owner_counts(Xd, None, B)
        for i in range(warmup):
            step(i)
        eng.flush_lazy(sync=False)
        torch.cuda.synchronize()

    best = None
    for rep_ in range(3):
        prepare_pass()
        t0 = time.perf_counter()
        for i in range(warmup, n_steps):
            step(i)
        eng.flush_lazy(sync=False)
        t_host = (time.perf_counter() - t0) / steps * 1e3
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        if best is None or ms < best[0]:
            best = (ms, t_host)
    prepare_pass()                      # an extra, untimed pass for the phases (recorded events on five steps)
    for i in range(warmup, warmup + 8):
        eng.timers = {} if i == warmup + 2 else eng.timers
        step(i)
    ph = eng.phase_ms()
    eng.timers = None
    best = (best[0], best[1], ph)
    ms, t_host, ph = best
    print(f"owner form, N={N}: {ms:.3f} ms/step per rank, fastest of 3 passes (host enqueue {t_host:.3f}; no communication; the slice "
          f"flush of the {steps} steps included) -> {N * B / ms / 1e3:.2f} M samples/s if comm were free; phases (median of 5 steps): " +
          ", ".join(f"{k} {v:.3f}" for k, v in ph.items()))


def run(N, steps=12, warmup=4, B=8192):
    os.environ["SATRANS_DP_MODE"] = "replicated"
    model = bench.build_model("cpu", 0.005)
    model.to("cuda:0"); model.device = "cuda:0"
    eng = model._require_engine()
    model.train()
    X, y = bench.synth_batches((steps + warmup) * B, 100)
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    others = []
    if N > 1:
        Xo, _ = bench.synth_batches((steps + warmup) * B * (N - 1), 777)
        Xo = torch.from_numpy(Xo).cuda().long()
        big = eng.row_span[:, 0] >= eng.small_rows                                  # large-table fields only
        off = eng.row_span[big, 0][None, :]
        rows_o = (Xo[:, eng.cols.long()[big]] + off).to(torch.int32).reshape(steps + warmup, -1)
        others = [rows_o[i].contiguous() for i in range(steps + warmup)]
    state = {"i": 0}
    parallel.world_size = lambda: N
    parallel.exchange_enabled = lambda: N > 1          # the step's multi-rank branch, fed by the stand-ins below
    parallel.gather_rows = lambda rows: torch.cat([rows.reshape(-1), others[state["i"]]]) if N > 1 else rows.reshape(-1)
    parallel.gather_grad_rows_async = lambda g: ((g.reshape(-1, g.shape[-1]).repeat(N, 1) if N > 1 else g.reshape(-1, g.shape[-1])), None)
    parallel.all_reduce_flat = lambda flat: None
    eng.timers = None
    for i in range(warmup):
        state["i"] = i
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        state["i"] = i
        eng.timers = {} if i == warmup + steps - 1 else None
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B])
    ph = eng.phase_ms()
    eng.timers = None
    eng.flush_lazy()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"N={N}: {ms:.3f} ms/step per rank (no communication) -> {N * B / ms / 1e3:.2f} M samples/s if comm were free;"
          f" phases of the last step: " + ", ".join(f"{k} {v:.3f}" for k, v in ph.items()))


if __name__ == "__main__":
    args = sys.argv[1:]
    mode = "owner"
    if args and args[0] == "--mode":
        mode, args = args[1], args[2:]
    ns = [int(a) for a in args] or [1, 2, 4, 8]
    if len(ns) > 1:
        # one PROCESS per rank count (this parent never touches the GPU): a second engine in the same process gets HIP streams
        # that share hardware queues with the first one's leftovers - every second run came out 30 % slower whatever its N
        import subprocess
        for n in ns:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--mode", mode, str(n)], check=False)
    else:
        (run_owner if mode == "owner" else run)(ns[0])
