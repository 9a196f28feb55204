"""The data-parallel training step with row OWNERSHIP (the default form with several ranks): every rank owns a
contiguous slice of the large tables' rows - values, both Adam moments, the lazy form's `last_step` - and is the only rank that
reads or writes them during training.  Mixed into `engine.PathEngine`, which holds the state these methods use (workspaces,
optimizer tensors, helper streams) and the parts shared with the other step forms (backward, next-batch preparation, flush).
The one-rank step is `engine_local.py`, the replicated-update form `engine_replicated.py`; `PathEngine.train_step` picks.
Gate: the two-ranks-on-one-GPU tests and the one-rank-through-RCCL test (tests/test_gpu_parity.py)."""
from __future__ import annotations

import contextlib
import math
import ctypes as C
from typing import List, Optional

import torch

from . import native as N


class OwnerStepMixin:
    def _owner_ranges(self, world: int, B: Optional[int] = None) -> List[int]:
        """Arena row boundaries of the owners' slices of the LARGE tables: rank o steps rows [b[o], b[o+1]).  (Small tables sit
        first in the arena, are stepped densely by every rank from the all-reduced gradient and have no owner.)

        The slices carry equal WORK, not equal row counts.  Every field sends B positions per rank and step into its table
        whatever the table's size (AliCCP: one table holds 66 % of the large rows and receives 1/8 of the positions), and an
        owner pays per received position (sort, replay, ordered sums, Adam: ~2.3 ns) and per owned row (the postponed
        regulariser steps at flush time: ~0.026 ns per row and step).  With uniform ids a row of a table with R rows used by k
        fields costs  90 * k * N B / R + 1  row-flush units per step (N ranks); the boundaries cut the cumulative cost into equal parts.
        Fixed at the first owner-form step (ownership of optimizer state must not move with the batch size) until the replicas
        are brought together again.  A pure function of the model, `world` and that first B: identical on every rank."""
        cached = getattr(self, "_owner_bounds", None)
        if cached is not None and cached[0] == world:
            return cached[1]
        lo0 = self.small_rows
        spans = sorted({(int(lo), int(hi)) for lo, hi in self.row_span.tolist() if lo >= lo0})
        uses = {sp: sum(1 for lo, hi in self.row_span.tolist() if (int(lo), int(hi)) == sp) for sp in spans}
        Bn = float(B or 8192) * world                                                      # positions per field and step, all ranks
        dens = [90.0 * uses[sp] * Bn / max(1, sp[1] - sp[0]) + 1.0 for sp in spans]       # cost units per row
        total = sum(d * (sp[1] - sp[0]) for d, sp in zip(dens, spans))
        bounds, acc, k = [lo0], 0.0, 1
        for d, (lo, hi) in zip(dens, spans):
            cost = d * (hi - lo)
            while k < world and acc + cost >= total * k / world:
                bounds.append(min(hi, lo + int(math.ceil((total * k / world - acc) / d))))
                k += 1
            acc += cost
        while len(bounds) < world:
            bounds.append(self.total_rows)
        bounds.append(self.total_rows)
        for i in range(1, len(bounds)):
            bounds[i] = max(bounds[i], bounds[i - 1])
        if B is not None:
            self._owner_bounds = (world, bounds)
        return bounds

    def plans_owner_counts(self) -> bool:
        """Whether plan_owner_counts would plan anything (fit() only waits for the whole sample order when it does)."""
        from . import parallel
        return bool(parallel.exchange_enabled() and self.dp_mode == "owner" and self.lazy and self.F_small < self.F)

    def plan_owner_counts(self, ids: torch.Tensor, order: Optional[torch.Tensor], batch_size: int) -> None:
        """Owner form, optional: the per-step all-to-all split sizes of a whole epoch in ONE pass and ONE read-back, for callers
        that know the epoch's batches ahead (`fit`: the resident id matrix [N, C] and the epoch's sample order).  Without a plan
        every step gathers its counts and reads them back before it can size its exchange, which drains the launch queue once
        per step.  Consumed step by step by train_step; a batch whose size does not match the plan falls back to the read-back."""
        from . import parallel
        self._owner_plan = None
        world = parallel.world_size()
        if not self.plans_owner_counts():
            return
        if ids.dtype not in (torch.float32, torch.int32, torch.int64) or ids.dim() != 2:
            return
        big = (self.row_span[:, 0] >= self.small_rows).nonzero().reshape(-1)
        cols = self.cols.long()[big]
        if self._owner_world != world:                 # (the plan and the steps must cut the rows at the same places)
            self.flush_lazy()
            self._owner_bounds = None
            self._owner_world = world
        if self.owner_prefetch:
            parallel.prefetch_group()                  # (collective, idempotent: every rank plans its epoch here)
        inner = torch.tensor(self._owner_ranges(world, batch_size)[1:-1], dtype=torch.int64, device=self.dev)
        n = ids.shape[0]
        steps = (n - 1) // batch_size + 1
        per = torch.zeros(steps, world, dtype=torch.int64, device=self.dev)
        step_of = torch.arange(n, device=self.dev) // batch_size                       # position in the epoch -> step
        for lo in range(0, n, 1 << 22):                                                # (bounded temporaries on 42 M-row datasets)
            hi = min(n, lo + (1 << 22))
            sel = order[lo:hi] if order is not None else slice(lo, hi)
            idv = ids[sel][:, cols].long()
            size = (self.row_span[big, 1] - self.row_span[big, 0])[None, :]
            # an id outside its table is recorded as the table's FIRST row by the gather kernel (and flagged: the step raises
            # IndexError afterwards); the plan must count it where the step will send it, or the all-to-all sizes disagree
            idv = torch.where((idv < 0) | (idv >= size), torch.zeros_like(idv), idv)
            rows = idv + self.row_span[big, 0][None, :]
            owner = torch.bucketize(rows, inner, right=True)                              # rows >= a boundary belong to the next owner
            key = step_of[lo:hi, None] * world + owner
            per.view(-1).scatter_add_(0, key.reshape(-1), torch.ones_like(key.reshape(-1)))
        allc = torch.empty(world * per.numel(), dtype=torch.int64, device=self.dev)
        parallel._all_gather(allc, per.reshape(-1))
        self._owner_plan = dict(counts=allc.reshape(world, steps, world).permute(1, 0, 2).contiguous().cpu(), step=0,
                                batch=batch_size, last=n - (steps - 1) * batch_size)

    def _owner_ws(self, ws: dict, half: int, n_b: int, need: int) -> dict:
        """Buffers for the lists an owner receives, one set per half of the next-batch double buffer.  Their length depends on the
        ids of the step (about n_b with uniform ids, up to world * n_b when every rank gathers from one slice): capacity doubles
        when a step outgrows it."""
        halves = ws.setdefault("_owner", [None, None])
        ow = halves[half]
        if ow is not None and ow["cap"] >= need:
            return ow
        lib, D, dev = self.lib, self.D, self.dev
        cap = max(2 * n_b, 1024, 1 << max(need - 1, 1).bit_length())
        i32 = dict(dtype=torch.int32, device=dev)
        n_reg = int(lib.satrans_embed_reg_partials(self.total_rows, cap, D))
        ow = dict(cap=cap, n_reg=n_reg,
                  ids=torch.empty(cap, **i32),                                     # the row ids as received (rank-major)
                  sorted=torch.empty(cap, **i32), src=torch.empty(cap, **i32), iota=torch.arange(cap, **i32),
                  sort_ws=torch.empty(int(lib.satrans_embed_sort_workspace_bytes(cap, self.total_rows)), dtype=torch.uint8, device=dev),
                  partial_ws=torch.empty(int(lib.satrans_embed_partial_ws_floats(cap, D)), dtype=torch.float32, device=dev),
                  vals=torch.empty(cap, D, dtype=torch.float32, device=dev),       # values out / gradient rows in
                  reg=torch.zeros(n_reg + (cap * D + 255) // 256, dtype=torch.float64, device=dev),
                  inv=(ow or {}).get("inv"))                                       # [B F]: (b, f) -> its row of the sorted value buffer
        halves[half] = ow
        return ow

    def _owner_counts(self, B: int, big_sorted, n_b: int, world: int, bt, peek: bool):
        """[N, N] split sizes of this step's (peek: the NEXT step's) row-id exchange: from the epoch plan (plan_owner_counts: no
        read-back, the launch queue stays full), else - not for a peek - gathered and read back now."""
        from . import parallel
        plan = getattr(self, "_owner_plan", None)
        if plan is not None:
            i = plan["step"]
            steps = plan["counts"].shape[0]
            if i < steps and B == (plan["last"] if i == steps - 1 else plan["batch"]):
                if not peek:
                    plan["step"] = i + 1
                return plan["counts"][i]
            if not peek:
                self._owner_plan = None
        if peek:
            return None
        cut = torch.searchsorted(big_sorted, bt[1]) if world > 1 else big_sorted.new_zeros(0, dtype=torch.int64)
        edges = torch.cat([cut.new_zeros(1), cut, cut.new_full((1,), n_b)])
        return parallel.gather_counts(edges[1:] - edges[:-1])        # [N, N] on the host: the step's one read-back

    def _owner_exchange_ids(self, ws, half, bufs, counts, B, group=None):
        """Everything of an owner-form step that depends on nothing but the batch's ids: the large-table part of the sorted rows
        to their owners (all-to-all, int32), the owner-side sort of what arrived (rank-major, then position: the order of the
        replicated form, hence its bits), the inverse of the batch's own sort (token (b, f) -> its row of the sorted value
        buffer).  On the CURRENT stream: the launch stream at the top of a step, or - the next batch's - the side stream under
        the previous step's tail (_prepare_owner_async).  `bufs`: the half of the double buffer holding the batch's rows."""
        from . import parallel
        lib, st = self.lib, self._stream()
        rank = parallel.rank()
        n_loc, n_s = B * self.F, B * self.F_small
        n_b = n_loc - n_s
        send, recv = counts[rank].tolist(), counts[:, rank].tolist()
        n_recv = int(sum(recv))
        ow = self._owner_ws(ws, half, n_b, max(n_recv, 1))
        ow["reg"].zero_()                                                        # (slot counts follow n_recv: no stale partials)
        big_sorted = bufs["sorted_rows"][n_s:]
        parallel.all_to_all_rows(big_sorted, send, recv, "all_to_all_row_ids_i32", out=ow["ids"], group=group)
        if n_recv:
            # what arrived is one sorted run per sending rank: ONE ranking launch merges them (the device-wide sort of the same
            # list: a block sort + ~10 merge passes, 48-61 us)
            starts = [0]
            for c in recv:
                starts.append(starts[-1] + int(c))
            if len(recv) <= 64:
                N.check(lib.satrans_embed_merge_runs(ow["ids"].data_ptr(), n_recv, (C.c_int64 * len(starts))(*starts), len(recv),
                                                     ow["sorted"].data_ptr(), ow["src"].data_ptr(), st), "satrans_embed_merge_runs")
            else:
                N.check(lib.satrans_embed_sort(ow["ids"].data_ptr(), n_recv, self.total_rows, ow["sorted"].data_ptr(),
                                               ow["src"].data_ptr(), None, ow["sort_ws"].data_ptr(), ow["sort_ws"].numel(),
                                               ow["iota"].data_ptr(), st), "satrans_embed_sort(owner)")
        if ow["inv"] is None or ow["inv"].numel() != n_loc:
            ow["inv"] = torch.empty(n_loc, dtype=torch.int32, device=self.dev)
        N.check(lib.satrans_embed_inverse_positions(bufs["src"].data_ptr(), n_loc, ow["inv"].data_ptr(), st),
                "satrans_embed_inverse_positions")
        return dict(ow=ow, send=send, recv=recv, n_recv=n_recv, B=B, half=half)

    def _prepare_owner_async(self, X_next, world, bt, after=None):
        """The id exchange of the NEXT step on the side stream, behind that batch's sort (_prepare_async), on a process group of
        its own (parallel.prefetch_group) - issued by the host AFTER this step's gradient collectives, so that every rank issues
        its collectives in the same order.  Only with an epoch plan (the split sizes are then known without a read-back)."""
        from . import parallel
        prep = self._prep
        if prep is None or prep.get("owner") is not None or self._side is None:
            return
        group = parallel._PREFETCH_GROUP               # (read only: created in front of the step, _train_step_owner)
        if not self.owner_prefetch or group is None:
            if prep["done"] is None:                   # no exchange ahead: only the deferred bucketing is left to do
                with torch.cuda.stream(self._side):
                    self._bucket(prep["X"], self.train_workspace(prep["B"], 1, False)["prep_alt"])
                    prep["done"] = torch.cuda.Event()
                    prep["done"].record(self._side)
            return
        B = prep["B"]
        ws = self.train_workspace(B, 1, False)        # (the NEXT batch's workspace: a ragged last batch has one of its own)
        n_b = B * (self.F - self.F_small)
        counts = self._owner_counts(B, None, n_b, world, bt, peek=True)
        with torch.cuda.stream(self._side):
            if counts is not None:
                half = 1 - ws.get("_own_half", 0)
                # Two communicators: RCCL kernels of different communicators must reach the device in the same order on every
                # rank, and host issue order alone does not give that (the side stream starts where a rank-local event fires).
                # The exchange therefore waits for THIS step's gradient collectives (events behind the gradient-row all-to-all on
                # the launch stream and behind the all-reduce on the tail stream): on every rank the default group's collectives
                # of step t are complete before the prefetch group's collective of step t + 1 can start, and the next step's
                # first default-group collective (the row answer) waits for this exchange - one total order, never two
                # communicators' kernels in flight at once.
                for ev in after or ():
                    self._side.wait_event(ev)
                with self.phase("owner_ids_next"):
                    own = self._owner_exchange_ids(ws, half, ws["prep_alt"], counts, B, group=group)
                own["done"] = torch.cuda.Event()
                own["done"].record(self._side)
                prep["owner"] = own
            if prep["done"] is None:                   # the bucketing, deferred behind the exchange (_prepare_async)
                self._bucket(prep["X"], ws["prep_alt"])
                prep["done"] = torch.cuda.Event()
                prep["done"].record(self._side)

    def _train_step_owner(self, X, y, B, world, ws, next_X=None):
        """One optimizer step of every data-parallel rank with row OWNERSHIP (reference semantics unchanged: per-GPU batches,
        loss summed over all samples, one dense Adam + L2 step, meta_basemodel.py:272-275,317; main.py:343).

        Rank o owns a contiguous 1/N slice of the large tables' rows - values, both Adam moments and the lazy form's `last` -
        and is the only rank that reads or writes them during training:
          ids      every rank sorts its batch's rows; the large-table part splits into N runs by owner   all-to-all (int32)
                   -> the owner sorts what it received          [both a step AHEAD, on the side stream: _prepare_owner_async]
          values   the owner replays the postponed steps of the requested rows, reads them               all-to-all back (fp32)
                   -> layer 0 reads its tokens from the received rows (plus the replicated small tables)
          grads    gradient rows of the large tables, packed in sorted order                             all-to-all (fp32)
                   -> ordered segmented sums over the owner's sorted list (rank-major, then position: the order of the
                      replicated form, hence the same bits), Adam on its slice
          small tables and dense parameters: one SUM all-reduce of the flat gradient buffer, dense step on every rank
        Per rank and step: ~n_b rows received and stepped, 2 x n_b x D x 4 bytes moved each way (8.4 MB at B = 8192) instead
        of N x n_b rows sorted, replayed and stepped and N x 8.4 MB received; the flush of the postponed steps covers 1/N of
        the rows.  The replicas of rows a rank does not own go stale; flush_lazy() brings them together again (slice broadcasts)
        before anything reads the tables as a whole.

        What sits on the launch stream between two steps' layer kernels: replay -> pack values -> all-to-all -> [layers] ->
        pack gradient rows -> all-to-all -> touched-row Adam.  Beside it, on the tail stream: slab reduction + scenario-table
        backward + small-table sums -> all-reduce -> dense step of the small tables + flat Adam; on the side stream: the next
        batch's sort, bucketing, id exchange and owner-side sort."""
        from . import parallel
        lib, m, D, st = self.lib, self.m, self.D, self._stream()
        n_loc, n_s = B * self.F, B * self.F_small
        n_b = n_loc - n_s
        l2 = m.l2_reg_embedding
        arena_t, am_t, av_t = m.embedding_arena, self.adam_m, self.adam_v
        arena, am, av = arena_t.data_ptr(), am_t.data_ptr(), av_t.data_ptr()
        if self._owner_world != world:
            self.flush_lazy()                         # (first owner-form step: everything current and identical everywhere)
            self._owner_bounds = None
            self._owner_world = world
        if self.owner_prefetch and parallel._PREFETCH_GROUP is None:
            # created and warmed HERE, in front of every collective of the step and with nothing in flight (collective: every
            # rank takes its first owner-form step at the same point of the program) - never lazily under the side stream
            parallel.prefetch_group()
        bounds = self._owner_ranges(world, B)
        if "xg" not in ws:
            ws["xg"] = torch.empty(n_loc, D, dtype=torch.float32, device=self.dev)
            ws["packed_o"] = torch.empty(max(n_b, 1), D, dtype=torch.float32, device=self.dev)
        bt = getattr(self, "_owner_bounds_t", None)
        if bt is None or bt[0] is not bounds:
            bt = self._owner_bounds_t = (bounds, torch.tensor(bounds[1:-1], dtype=torch.int32, device=self.dev))
        main = torch.cuda.current_stream(self.dev)

        # ---- 1. this batch's arena rows, sorted, and their exchange with the owners - prepared by the previous step on the
        #         side stream (_prepare_async + _prepare_owner_async), else here ---------------------------------------------
        own = (self._prep or {}).get("owner")
        prepared = self._take_prepared(X, ws)
        if not prepared:
            if own is not None:
                main.wait_event(own["done"])      # (the discarded exchange still owns its half of the buffers)
                if world > 1:
                    # The id exchange of the batch named with `next_X` is already done - by EVERY rank, on the prefetch group.
                    # Replacing it here means one more default-group all-to-all that only the ranks whose hint was wrong would
                    # enter: a hang, not a slow path.  The decision cannot be rank-local, so it is an error.
                    raise RuntimeError("train_step (owner form, several ranks): this batch is not the one the previous step named "
                                       "with next_X, and that batch's id exchange has already run on every rank.  Pass the tensor "
                                       "of the next call as next_X on every rank (fit does), or no next_X at all, or set "
                                       "SATRANS_OWNER_PREFETCH=0.")
            own = None
            N.check(lib.satrans_gather_fwd(arena, self.row_span.data_ptr(), self.cols.data_ptr(), X.data_ptr(),
                                           N.id_dtype_of(X), X.stride(0), B, self.F, D, None, ws["rows"].data_ptr(),
                                           self.status.data_ptr(), st), "satrans_gather_fwd(rows)")
            with self.phase("embed_sort"):
                if self._sort_fields is not None and B <= 8192:
                    f_, lo_, n_ = self._sort_fields
                    N.check(lib.satrans_embed_sort_fields(ws["rows"].data_ptr(), B, self.F, f_, lo_, n_, ws["sorted_rows"].data_ptr(),
                                                          ws["src"].data_ptr(), st), "satrans_embed_sort_fields")
                else:
                    N.check(lib.satrans_embed_sort(ws["rows"].data_ptr(), n_loc, self.total_rows, ws["sorted_rows"].data_ptr(),
                                                   ws["src"].data_ptr(), None, ws["sort_ws"].data_ptr(), ws["sort_ws"].numel(),
                                                   ws["iota"].data_ptr(), st), "satrans_embed_sort")
        self.adam_t += 1
        self._note_lr(m._adam_cfg["lr"])
        h_emb = self._hparams(l2)
        big_sorted, big_src = ws["sorted_rows"][n_s:], ws["src"][n_s:]
        if own is not None and own["B"] == B:
            ws["_own_half"] = own["half"]
            self._owner_counts(B, big_sorted, n_b, world, bt, peek=False)      # (the plan moves on: these are the counts `own` used)
            main.wait_event(own["done"])
        else:
            with self.phase("owner_ids"):
                counts = self._owner_counts(B, big_sorted, n_b, world, bt, peek=False)
                half = ws.get("_own_half", 0)
                own = self._owner_exchange_ids(ws, half, ws, counts, B)
        ow, send, recv, n_recv = own["ow"], own["send"], own["recv"], own["n_recv"]
        # ---- 2. the owner's side: replay the postponed steps of exactly the rows that were asked for, answer ----------------
        if n_recv and self.adam_t > 1:
            with self.phase("lazy_replay"):
                table, h = self._table(self.adam_t), self._hparams(l2)
                N.check(lib.satrans_embed_lazy_replay(arena, am, av, self.last_step.data_ptr(), D, ow["sorted"].data_ptr(),
                                                      n_recv, self.adam_t - 1, table.data_ptr(), C.byref(h),
                                                      ow["reg"][ow["n_reg"]:].data_ptr(), st), "satrans_embed_lazy_replay")
        with self.phase("owner_rows"):
            # the batch's rows in sorted order: replicated small tables from the arena, large tables from their owners (straight
            # into their part of the buffer); token (b, f) is row inv[b, f] of that buffer
            xg, inv = ws["xg"], ow["inv"]
            if n_recv:
                N.check(lib.satrans_embed_pack_rows(ow["ids"].data_ptr(), n_recv, arena, D, ow["vals"].data_ptr(), st),
                        "satrans_embed_pack_rows(values)")
            parallel.all_to_all_rows(ow["vals"][:n_recv], recv, send, "all_to_all_rows_f32", out=xg[n_s:])
            if n_s:
                self._join_flat()          # (the previous step's dense step of the small tables ran on the tail stream)
                N.check(lib.satrans_embed_pack_rows(ws["sorted_rows"].data_ptr(), n_s, arena, D, xg.data_ptr(), st),
                        "satrans_embed_pack_rows(small)")
        # ---- 3. forward, loss, backward on the received rows; the slab reduction and the scenario-table backward go to the tail
        #         stream (backward(side_tail=True)), the next batch's sort and bucketing to the side stream ----------------------
        self._x_src = (xg, inv)
        hook = None
        if next_X is not None and self._dense_override is None and self._can_prepare(next_X, next_X.shape[0]) \
                and next_X.shape[1] >= self.n_cols:
            hook = lambda fork: self._prepare_async(next_X, B, fork, defer_bucket=True)
        try:
            gemb = self.backward(X, y, ws, rows_ready=True, bucket_ready=prepared, after_layers=hook, side_tail=self.side_tail)
        finally:
            self._x_src = None
        tail = self._side_tail if self._tail_done is not None else None
        self._tail_done = None
        # ---- 4. large tables, launch stream: gradient rows to their owners, ordered sums + Adam on the owner's slice ------------
        with self.phase("owner_grads"):
            if n_b:
                N.check(lib.satrans_embed_pack_rows(big_src.data_ptr(), n_b, gemb.data_ptr(), D, ws["packed_o"].data_ptr(), st),
                        "satrans_embed_pack_rows(grads)")
            recv_g = parallel.all_to_all_rows(ws["packed_o"][:n_b], send, recv, "all_to_all_grad_rows_f32", out=ow["vals"])
        ev_grads = torch.cuda.Event()
        ev_grads.record(main)                         # (the launch stream has waited for the collective: it is complete here)
        # ---- 5. small tables + dense parameters, tail stream (beside 4): ordered sums, ONE all-reduce, dense steps --------------
        with (torch.cuda.stream(tail) if tail is not None else contextlib.nullcontext()):
            st_t = self._stream()
            if n_s > 0:
                with self.phase("adam_small"):
                    N.check(lib.satrans_embed_segment_sums(ws["sorted_rows"].data_ptr(), ws["src"].data_ptr(), n_s, gemb.data_ptr(), D,
                                                           ws["partial_ws"].data_ptr(), ws["reg_unused"].data_ptr(),
                                                           self.g_small.data_ptr(), st_t), "satrans_embed_segment_sums")
            parallel.all_reduce_flat(self.g_exchange)
            ev_reduce = torch.cuda.Event()
            ev_reduce.record(torch.cuda.current_stream(self.dev))
            if self.small_rows > 0:
                with self.phase("adam_small"):
                    N.check(lib.satrans_embed_adam_rows(arena, am, av, self.last_step.data_ptr(), 0, self.small_rows, D,
                                                        self.g_small.data_ptr(), C.byref(h_emb), self.adam_t,
                                                        ws["reg_rows"].data_ptr(), st_t), "satrans_embed_adam_rows")
                    N.check(lib.satrans_sum_f64(ws["reg_rows"].data_ptr(), ws["reg_rows"].numel(), self.reg_small.data_ptr(), 1,
                                                st_t), "satrans_sum_f64")
            h_flat = self._hparams(0.0, tables=False)
            with self.phase("adam_flat"):
                N.check(lib.satrans_adam_flat(m.flat_params.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                              self.flat_v.data_ptr(), m.flat_params.numel(), C.byref(h_flat), st_t),
                        "satrans_adam_flat")
            if tail is not None:
                self._flat_done = torch.cuda.Event()
                self._flat_done.record(tail)
                if self.preclear and next_X is not None:
                    self.flat_g_all.zero_()      # the NEXT step's gradient clear (pipelined callers only: train_step's docstring)
                    self._precleared = (torch.cuda.Event(), self.flat_g_all)
                    self._precleared[0].record(tail)
        # (the next batch's id exchange: issued behind this step's two gradient collectives - the same order on every rank)
        if hook is not None:
            self._prepare_owner_async(next_X, world, bt, after=(ev_grads, ev_reduce))
        if n_recv:
            with self.phase("adam_touched"):
                N.check(lib.satrans_embed_adam_touched(arena, am, av, D, ow["sorted"].data_ptr(), ow["src"].data_ptr(), n_recv,
                                                       recv_g.data_ptr(), ow["partial_ws"].data_ptr(), C.byref(h_emb),
                                                       ow["reg"].data_ptr(), self.last_step.data_ptr(), self.adam_t, st),
                        "satrans_embed_adam_touched")
        N.check(lib.satrans_sum_f64(ow["reg"].data_ptr(), ow["reg"].numel(), self.reg_sum.data_ptr(), 1, st), "satrans_sum_f64")
        self._lazy_pending = True
        self._replicas_stale = True
        self._since_flush += 1
        self._stepped_since_forward = True
        if tail is None or next_X is None:
            self._join_flat()              # (a caller outside a fit-style loop sees the finished step, as with any torch op)
        if self.flush_every and self._since_flush >= self.flush_every:
            self.flush_lazy(sync=False)               # own slice only: nobody reads the other replicas during training

    def _sync_replicas(self):
        """Owner form: every rank receives the other owners' slices (values, both moments) - after this the replicas are
        identical everywhere and every row is at the current step."""
        from . import parallel
        if self._owner_world:
            b = self._owner_ranges(self._owner_world)
            with self.phase("sync_replicas"):
                for o in range(self._owner_world):
                    if b[o + 1] > b[o]:
                        for t in (self.m.embedding_arena, self.adam_m, self.adam_v):
                            parallel.broadcast_slice(t[b[o]:b[o + 1]], o)
                self.last_step.fill_(self.adam_t)
        self._replicas_stale = False

