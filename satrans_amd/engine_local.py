"""The training step of ONE rank: everything on the launch stream except the tail (slab reduction, scenario-table
backward, flat Adam: tail stream) and the next batch's preparation (side stream).  Mixed into `engine.PathEngine`; the
several-rank forms are `engine_owner.py` (row ownership, the default) and `engine_replicated.py`."""
from __future__ import annotations

import contextlib
import ctypes as C

import torch

from . import native as N
from .streams import shared_stream


class LocalStepMixin:
    def _train_step_local(self, X, y, B, cfg, next_X=None):
        """ids -> arena rows + per-field sort (or taken from the previous step's preparation) -> [lazy: replay of the rows this
        batch reads] -> [streaming form: every other row's regulariser-only step on a side stream underneath the forward] ->
        forward, loss, backward -> ordered segmented sums + Adam over the sorted (row, gradient) list -> flat Adam."""
        ws = self.train_workspace(B, 1, False)
        lib, m, D = self.lib, self.m, self.D
        main = torch.cuda.current_stream(self.dev)
        st = self._stream()
        # table classes only pay off when there is an exchange to shrink (one rank: +5 launches for nothing; tests force them)
        split = self.force_split
        small_rows = self.small_rows if split else 0
        n_loc = B * self.F
        n_s = B * self.F_small if split else 0
        n_b = n_loc - n_s
        l2 = m.l2_reg_embedding
        arena, am, av = m.embedding_arena.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr()

        # ---- 1. this batch's arena rows (nothing is moved yet), sorted - unless the previous step prepared them already ----
        prepared = self._take_prepared(X, ws)
        if not prepared:
            self._rows_sorted(X, ws, B, st)
        self.adam_t += 1
        self._note_lr(cfg["lr"])
        h_emb = self._hparams(l2)
        # ---- 2. lazy form: replay the postponed steps of exactly these rows up to t-1, so that the gather reads current
        #         values (small-table rows are always current: they take a dense step every step) --------------------------
        self._join_roll()
        if self.lazy and self.adam_t > 1:
            with self.phase("lazy_replay"):
                self._replay_rows(ws["sorted_rows"], n_loc, ws["replay_reg"], None)
        elif self.lazy:
            ws["replay_reg"].zero_()
        rolling = self.rolling_flush if self.lazy and self.flush_every > 1 and not split else ""
        rolled = False
        if rolling == "step":
            fork = torch.cuda.Event()
            fork.record(main)
            rolled = self._roll_flush(fork)
        # ---- 3. streaming form of the dense step: every other row takes its regulariser-only step now, on a side stream
        #         underneath the forward ---------------------------------------------------------------------------------------
        side_done = None
        big_sorted, big_src = ws["sorted_rows"][n_s:], ws["src"][n_s:]
        if not self.lazy:
            use_side = self.overlap
            if use_side:
                if self._side is None:
                    self._side = shared_stream(self.dev, "side", lambda: torch.cuda.Stream(self.dev))
                ready = torch.cuda.Event()
                ready.record(main)
                self._side.wait_event(ready)
            with torch.cuda.stream(self._side) if use_side else contextlib.nullcontext():
                N.check(lib.satrans_embed_mark_touched(big_sorted.data_ptr() if n_b else None, n_b, self.total_rows,
                                                       ws["touched"].data_ptr(), self._stream()),
                        "satrans_embed_mark_touched")
                if self.total_rows > small_rows:
                    with self.phase("adam_untouched"):
                        N.check(lib.satrans_embed_adam_untouched(arena, am, av, small_rows, self.total_rows, D,
                                                                 ws["touched"].data_ptr(), C.byref(h_emb),
                                                                 ws["reg_partials"].data_ptr(), 0, self._stream()),
                                "satrans_embed_adam_untouched")
                if use_side:
                    side_done = torch.cuda.Event()
                    side_done.record(self._side)

        # ---- 4. forward, loss, backward ---------------------------------------------------------------------------------
        hook = None
        if next_X is not None and self._dense_override is None and self._can_prepare(next_X, next_X.shape[0]) \
                and next_X.shape[1] >= self.n_cols:
            hook = lambda fork: self._prepare_async(next_X, B, fork)
        gemb = self.backward(X, y, ws, rows_ready=True, bucket_ready=prepared, after_layers=hook,
                             side_tail=self.side_tail and not split)
        if rolling and rolling != "step":
            rolled = self._roll_flush(self._last_fork)

        # ---- 5. small tables (forced table classes only): ordered segmented sums into their dense gradient, dense step ----------
        if n_s > 0:
            with self.phase("adam_small"):
                N.check(lib.satrans_embed_segment_sums(ws["sorted_rows"].data_ptr(), ws["src"].data_ptr(), n_s,
                                                       gemb.data_ptr(), D, ws["partial_ws"].data_ptr(),
                                                       ws["reg_unused"].data_ptr(), self.g_small.data_ptr(), st),
                        "satrans_embed_segment_sums")
        if small_rows > 0:
            self._small_tables_step(ws, small_rows, h_emb, st)
        # ---- 6. large tables: the sorted (row, gradient row) list ---------------------------------------------------------------
        if side_done is not None:
            main.wait_event(side_done)
        if n_b > 0:
            with self.phase("adam_touched"):
                N.check(lib.satrans_embed_adam_touched(arena, am, av, D, big_sorted.data_ptr(), big_src.data_ptr(), n_b,
                                                       gemb.data_ptr(), ws["partial_ws"].data_ptr(), C.byref(h_emb),
                                                       ws["reg_partials"].data_ptr(),
                                                       self.last_step.data_ptr() if self.lazy else None, self.adam_t, st),
                        "satrans_embed_adam_touched")
        if self.lazy:
            self._lazy_pending = True
            self._since_flush += 1
        self._stepped_since_forward = True
        h_flat = self._hparams(0.0, tables=False)
        if self._tail_done is not None:
            # The dense gradients were finished on their own stream: the flat Adam launch follows them THERE (a join here would
            # cost the launch stream ~14 us of wake-up latency for an event that completes about when the touched-row chain
            # does), the launch stream only adds up the step's regulariser partial sums.  The next reader of the flat
            # parameters - the next step's scenario tables, behind its replay launch - waits for `_flat_done`, long complete.
            self._tail_done = None
            with torch.cuda.stream(self._side_tail):
                with self.phase("adam_flat"):
                    N.check(lib.satrans_adam_flat(m.flat_params.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                                  self.flat_v.data_ptr(), m.flat_params.numel(), C.byref(h_flat),
                                                  self._stream()), "satrans_adam_flat")
                self._flat_done = torch.cuda.Event()
                self._flat_done.record(self._side_tail)
                if self.preclear and not split and next_X is not None:
                    # the NEXT step's gradient clear, behind the event: 6 us and a launch gap off the head of that step's reduction
                    # (pipelined callers only: without `next_X` the gradients stay readable - ADVICE r04)
                    self._g_step_tail.zero_()
                    self._precleared = (torch.cuda.Event(), self._g_step_tail)
                    self._precleared[0].record(self._side_tail)
            # (the step's regulariser partial sums on that stream as well, behind an event of the touched-row chain and with two
            #  alternating sets of partial sums: 1.116-1.119 -> 1.120-1.122 ms/step, three A/B rounds - the sum is not what the next
            #  step's first launch waits for)
            N.check(lib.satrans_sum_f64(ws["reg_partials"].data_ptr(), ws["reg_partials"].numel(), self.reg_sum.data_ptr(), 1, st),
                    "satrans_sum_f64")
            if next_X is None:
                # no next step announced (a caller outside a fit-style loop): the launch stream joins right here, so that whatever
                # it does next - reading a parameter tensor directly, for instance - sees the finished step, as with any torch op.
                # A pipelined caller (fit, bench: `next_X` given) joins where the next step first reads the flat parameters.
                self._join_flat()
        else:
            with self.phase("adam_flat"):
                self._flat_step(h_flat, ws, st)
        if rolled:
            self._since_flush = 0          # (every row is at most flush_every steps behind by construction)
        elif self.lazy and self.flush_every and self._since_flush >= self.flush_every:
            self.flush_lazy()
