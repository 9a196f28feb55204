"""The data-parallel training step with REPLICATED updates (round 2's form; SATRANS_DP_MODE=replicated, kept under test): every
rank holds every table row and applies the same updates in the same order, so the replicas stay bit-identical (parallel.py).
Exchange per step: the large tables' row ids (all-gather), the flat gradient + the dense gradient of the small tables (one SUM
all-reduce), the large tables' gradient rows (all-gather, asynchronous) - with the global sort and the replay of the other
ranks' rows underneath the last one.  Mixed into `engine.PathEngine`; the default several-rank form is `engine_owner.py`."""
from __future__ import annotations

import ctypes as C

from . import native as N


class ReplicatedStepMixin:
    def _train_step_replicated(self, X, y, B, world, cfg):
        from . import parallel
        ws = self.train_workspace(B, world, True)
        lib, m, D = self.lib, self.m, self.D
        st = self._stream()
        small_rows = self.small_rows
        n_loc = B * self.F
        n_s = B * self.F_small
        n_b = n_loc - n_s
        n_big = n_b * world
        l2 = m.l2_reg_embedding
        arena, am, av = m.embedding_arena.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr()

        # ---- 1. this batch's arena rows (nothing is moved yet), sorted ----------------------------------------------------------
        self._rows_sorted(X, ws, B, st)
        self.adam_t += 1
        self._note_lr(cfg["lr"])
        h_emb = self._hparams(l2)
        # ---- 2. lazy form: replay the postponed steps of exactly these rows up to t-1, so that the gather reads current values ----
        if self.lazy and self.adam_t > 1:
            with self.phase("lazy_replay"):
                self._replay_rows(ws["sorted_rows"], n_loc, ws["replay_reg"], None)
        elif self.lazy:
            ws["replay_reg"].zero_()
        big_sorted, big_src = ws["sorted_rows"][n_s:], ws["src"][n_s:]

        # ---- 3. forward, loss, backward ---------------------------------------------------------------------------------------
        gemb = self.backward(X, y, ws, rows_ready=True, bucket_ready=False, after_layers=None, side_tail=False)

        # ---- 4. small tables: ordered segmented sums into the dense gradient at the tail of the flat gradient buffer -----------
        if n_s > 0:
            with self.phase("adam_small"):
                N.check(lib.satrans_embed_segment_sums(ws["sorted_rows"].data_ptr(), ws["src"].data_ptr(), n_s,
                                                       gemb.data_ptr(), D, ws["partial_ws"].data_ptr(),
                                                       ws["reg_unused"].data_ptr(), self.g_small.data_ptr(), st),
                        "satrans_embed_segment_sums")
        # ---- 5. the exchange.  Collectives in this order: large-table row ids (small), flat gradient + small tables (small),
        #         large-table gradient rows (the big one, asynchronous).  While the rows travel the GPU has nothing else to do,
        #         so the work that only needs the ids runs now: the global sort and the replay of the other ranks' rows (lazy
        #         form) or the streaming step of every untouched row. --------------------------------------------------------------
        grads, pending = gemb, None
        all_rows = parallel.gather_rows(big_sorted) if n_b > 0 else None
        parallel.all_reduce_flat(self.g_exchange)
        if n_b > 0:
            N.check(lib.satrans_embed_pack_rows(ws["src"][n_s:].data_ptr(), n_b, gemb.data_ptr(), D,
                                                ws["packed"].data_ptr(), st), "satrans_embed_pack_rows")
            grads, pending = parallel.gather_grad_rows_async(ws["packed"])
            with self.phase("embed_sort_global"):
                self._sort_rows(ws, B, all_rows, n_big, ws["g_sorted"], ws["g_src"], None if self.lazy else ws["touched"].data_ptr())
            big_sorted, big_src = ws["g_sorted"], ws["g_src"]
            if self.lazy and self.adam_t > 1:
                with self.phase("lazy_replay_global"):
                    self._replay_rows(big_sorted, n_big, ws["replay_reg_g"], self.reg_sum)
        elif not self.lazy:
            N.check(lib.satrans_embed_mark_touched(None, 0, self.total_rows, ws["touched"].data_ptr(), st),
                    "satrans_embed_mark_touched")
        if not self.lazy and self.total_rows > small_rows:
            with self.phase("adam_untouched"):
                N.check(lib.satrans_embed_adam_untouched(arena, am, av, small_rows, self.total_rows, D,
                                                         ws["touched"].data_ptr(), C.byref(h_emb),
                                                         ws["reg_partials"].data_ptr(), 0, self._stream()),
                        "satrans_embed_adam_untouched")
        # ---- 6. small tables: dense step over all their rows ---------------------------------------------------------------------
        if small_rows > 0:
            self._small_tables_step(ws, small_rows, h_emb, st)
        # ---- 7. large tables: (row, gradient row) lists of all ranks ------------------------------------------------------------
        if pending is not None:
            pending.wait()
        if n_big > 0:
            with self.phase("adam_touched"):
                N.check(lib.satrans_embed_adam_touched(arena, am, av, D, big_sorted.data_ptr(), big_src.data_ptr(), n_big,
                                                       grads.data_ptr(), ws["partial_ws"].data_ptr(), C.byref(h_emb),
                                                       ws["reg_partials"].data_ptr(),
                                                       self.last_step.data_ptr() if self.lazy else None, self.adam_t, st),
                        "satrans_embed_adam_touched")
        if self.lazy:
            self._lazy_pending = True
            self._since_flush += 1
        self._stepped_since_forward = True
        h_flat = self._hparams(0.0, tables=False)
        self._tail_done = None                         # (no tail stream in this form: backward(side_tail=False) joined it)
        with self.phase("adam_flat"):
            self._flat_step(h_flat, ws, st)
        if self.lazy and self.flush_every and self._since_flush >= self.flush_every:
            self.flush_lazy()
