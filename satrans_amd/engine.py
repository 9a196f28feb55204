"""Host-side driver of the HIP kernels for one SATrans model instance.

Owns the per-batch-size workspaces, launches the kernels through the C ABI (satrans_amd.native) on torch's
current HIP stream, and keeps the optimizer state.  PyTorch is used here for device memory, streams and the
[S, P] scenario tables (a handful of tiny ops on S <= a few rows); every per-sample computation is a HIP kernel.

Step anatomy (one training step, one GPU):
    launch stream   replay of the batch's postponed row updates -> scenario tables -> L-1 x layer_fwd (layer 0 reads its tokens
                    from the arena: the gather fused in) -> last layer + head + loss + their backward (ONE launch) -> L-1 x
                    layer_bwd -> touched-row Adam chain -> regulariser partial sums
    tail stream     slab reduction of all layers -> scenario-table backward -> flat Adam -> next step's gradient clear
    side stream     (lowest priority) the NEXT batch's ids -> rows + per-field sort, scenario bucketing
With torch.distributed initialised (one process per GPU, RCCL) the step takes the row-ownership form (_train_step_owner, DESIGN.md
6): ids to the rows' owners a step ahead on the side stream, current rows back in front of layer 0, gradient rows to the owners
behind the last backward kernel, ONE all-reduce (SUM: the loss is a sum over samples) of the dense gradients on the tail stream.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
from typing import Dict, List, Optional

import torch

from . import native as N
from .inputs import split_columns


from .engine_local import LocalStepMixin
from .engine_owner import OwnerStepMixin
from .engine_replicated import ReplicatedStepMixin
from .streams import _STREAMS, shared_stream as _shared_stream      # (one set of helper streams per device and process)


class PathEngine(LocalStepMixin, ReplicatedStepMixin, OwnerStepMixin):
    def __init__(self, model):
        self.lib = N.lib()
        m = self.m = model
        self.dev = m.embedding_arena.device
        N.require_gpu(m.embedding_arena, "SATrans")
        flag = m.flag
        sparse, dense, _ = split_columns(m.dnn_feature_columns)
        self.F = len(sparse)
        self.D = m.embedding_size
        self.H = m.att_head_num
        self.L = m.domain_att_layer_num
        units = m.meta_dnn_hidden_units
        self.gate = 'gate' in flag                       # reference satrans.py:61-62 (tested before 'bilinear')
        self.bilinear = 'bilinear' in flag and not self.gate
        if 'bilinear' in flag and 'gate' in flag:
            # the reference applies the gate on Q/K AND the per-head bilinear map with a length-D vector reshaped to
            # [H,d,d], which only works when D == D*D/H; not a configuration anyone runs
            raise NotImplementedError("flags 'gate' and 'bilinear' together")
        self.metanet = not self.gate and not self.bilinear
        self.multi = len(m.domain_column_list) > 1
        if len(units) != 3 or units[2] != self.D:
            raise NotImplementedError(f"meta_dnn_hidden_units must be (U, embedding_dim); got {units[1:]}")
        self.U = units[1]
        self.P = m.meta_param_size
        self.S = m.domain_embeddings.weight.shape[0]
        if self.multi:
            # several scenario columns (satrans.py:205-207): the scenario embedding is the mean of those columns' rows,
            # so the "scenario" of a sample is the TUPLE of ids; it is indexed by a composite id over the product of the
            # columns' vocabularies and the generated-weight table gets one row per tuple
            if 'onlyemb' in flag:
                raise NotImplementedError("'onlyemb' with several scenario columns")
            tabs_ = [m.domain_embedding_dict[c.embedding_name] for c in m.domain_feature_columns]
            sizes = [t.weight.shape[0] for t in tabs_]
            self.S = 1
            for v in sizes:
                self.S *= v
            if self.S > 4096:
                raise NotImplementedError(f"{self.S} scenario tuples: composite scenario tables above 4096 rows")
            strides, acc = [], 1
            for v in reversed(sizes):
                strides.insert(0, acc)
                acc *= v
            comp = torch.arange(self.S, device=self.dev)
            self._multi_tables = tabs_
            self._multi_index = [(comp // st) % v for st, v in zip(strides, sizes)]
            self._multi_strides = torch.tensor(strides, dtype=torch.int64, device=self.dev)
        self.pos = 'pos' in flag
        self.onlyemb = 'onlyemb' in flag
        if self.onlyemb and self.pos:
            raise NotImplementedError("'onlyemb' together with 'pos' (the reference concatenates a width-P embedding with a "
                                      "width-D position vector there and fails in the layer)")
        # scenario tables: HIP kernels in every variant (csrc/scenario.hip).  plain = one scenario column, no positions: the
        # encoder reads the scenario embedding directly; otherwise its input rows E [LR*S, De] are assembled first.
        self.plain_tabs = not (self.multi or self.pos or self.onlyemb)
        self.LR = 2 * self.L if self.pos else 1
        De = (2 if self.pos else 1) * m.domain_embeddings.weight.shape[1]
        self.De = De
        if not self.onlyemb:
            self._tab_ws = torch.empty(int(self.lib.satrans_scenario_table_bwd_ws_floats(self.LR * self.S, De)),
                                       dtype=torch.float32, device=self.dev)
        if not self.plain_tabs and not self.onlyemb:
            self._E = torch.empty(self.LR * self.S, De, dtype=torch.float32, device=self.dev)
            self._gE = torch.empty_like(self._E)
            tabs_ = self._multi_tables if self.multi else [m.domain_embeddings]
            self._scen_tables = tabs_
            self._scen_ptrs = (C.c_void_p * len(tabs_))(*[t.weight.data_ptr() for t in tabs_])
            self._scen_rows = (C.c_int32 * len(tabs_))(*[int(t.weight.shape[0]) for t in tabs_])
            self._scen_index = torch.stack(self._multi_index).to(torch.int32).contiguous() if self.multi else None
        self.flags = 0
        if not self.bilinear and 'Q' in m.meta_mode:       # MetaNet or gate on the queries
            self.flags |= N.META_Q
        if not self.bilinear and 'K' in m.meta_mode:       # ... on the keys
            self.flags |= N.META_K
        if self.gate:
            self.flags |= N.GATE
        if self.bilinear:
            self.flags |= N.BILINEAR                        # per-head map of the queries, whatever meta_mode says
        if 'relu' in flag:
            self.flags |= N.RELU_OUT
        if not m.att_res:
            self.flags |= N.NO_RES
        self.drop_p = 0.1                                   # reference models/satrans.py:27-28

        fi = m.feature_index
        self.cols = torch.tensor([fi[c.name][0] for c in sparse], dtype=torch.int32, device=self.dev)
        dcols: List[int] = []
        for c in dense:
            dcols += list(range(fi[c.name][0], fi[c.name][1]))
        self.n_dense = len(dcols)
        self._dense_cols_host = list(dcols)
        self.dense_cols = torch.tensor(dcols, dtype=torch.int32, device=self.dev) if dcols else None
        self.dom_col = fi[m.domain_column_list[0]][0]
        if self.multi:
            self._multi_cols = torch.tensor([fi[c.name][0] for c in m.domain_feature_columns], dtype=torch.int64,
                                            device=self.dev)
        self.n_cols = max(e for _, e in fi.values())
        # arena rows [lo, hi) of every FIELD's table (fields sharing an embedding_name share a table)
        spans = [(m._table_rows[c.embedding_name][0], m._table_rows[c.embedding_name][0] + m._table_rows[c.embedding_name][1])
                 for c in sparse]
        self.total_rows = m.embedding_arena.shape[0]
        self.row_span = torch.tensor(spans, dtype=torch.int64, device=self.dev).contiguous()
        # optimizer classes (basemodel._rebind_storage puts the small tables first in the arena): arena row < small_rows
        # <=> small table.  Every sample contributes exactly one row per field, so after sorting a batch's rows the first
        # B * F_small positions are the small-table ones.
        self.small_rows = int(m._arena_small_rows)
        self.F_small = sum(1 for lo, _ in spans if lo < self.small_rows)
        # one-launch sort of a batch's rows, one workgroup per field (csrc/embed_adam.hip: satrans_embed_sort_fields): possible when
        # every field has a table of its own; the fields in arena order with the first row and the row count of their tables
        by_lo = sorted(range(len(spans)), key=lambda f: spans[f][0])
        own = all(spans[by_lo[k]][0] >= spans[by_lo[k - 1]][1] for k in range(1, len(by_lo))) and len(spans) <= 64
        self._sort_fields = None
        if own and os.environ.get("SATRANS_SORT_FIELDS", "1") != "0":
            Arr = C.c_int32 * len(spans)
            self._sort_fields = (Arr(*by_lo), Arr(*[spans[f][0] for f in by_lo]), Arr(*[spans[f][1] - spans[f][0] for f in by_lo]))

        self._ws: Dict[int, dict] = {}
        self.status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.loss_sum = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self.reg_sum = torch.zeros(1, dtype=torch.float64, device=self.dev)
        self.reg_roll = torch.zeros(1, dtype=torch.float64, device=self.dev)     # the rolling flush's share (its own stream)
        self.adam_t = 0
        self.adam_m = self.adam_v = None
        self.flat_m = self.flat_v = self.flat_g = None
        from . import parallel
        # every data-parallel rank draws different dropout masks
        self.drop_seed = int((torch.initial_seed() ^ (parallel.rank() * 0x9E3779B1)) & 0xFFFFFFFF)
        self.drop_step = 0
        self._last_prob = None
        # optional per-phase timing with HIP events recorded on the launch stream (bench.py): name -> [(start, end)]
        self.timers: Optional[Dict[str, list]] = None
        # streaming form only: the every-row step of untouched rows runs on a side stream under the layer kernels
        self.overlap = True
        self._side = None
        # Lazy-exact dense Adam (default): the regulariser-only steps of rows that are not gathered are postponed and
        # replayed - same arithmetic, same result bit for bit - when a row is next gathered, or for all rows by
        # flush_lazy() (epoch end, before predict / state_dict).  SATRANS_LAZY_ADAM=0: streaming kernel every step.
        self.lazy = os.environ.get("SATRANS_LAZY_ADAM", "1") != "0"
        # Lazy form: bring ALL rows up to date every `flush_every` steps (0 = only when something needs current tables: epoch
        # end, predict, state_dict).  The total arithmetic is the same either way (every row-step is executed exactly once,
        # at replay or at flush time), but a row that waits 1,000 steps for its next gather is replayed as ONE lane's chain of
        # 1,000 dependent steps inside the step that gathers it, while the flush runs the same steps with every lane busy.
        self.flush_every = int(os.environ.get("SATRANS_LAZY_FLUSH_EVERY", "32"))
        self._since_flush = 0
        # One rank, lazy form: instead of ONE launch over all rows every `flush_every` steps (2.6 ms at 6.57 M rows, on the launch
        # stream), every step brings 1/flush_every of the rows up to the PREVIOUS step on a lowest-priority stream forked behind
        # its last backward kernel (_roll_flush): the same row-steps, executed underneath the step's small tail kernels.
        # Values: "0" off; "tail" fork behind the step's last backward kernel; "step" fork behind the step's replay launch (the slice
        # has the whole step's layer kernels to hide under: pays where those are ordinary grids - the general path).
        # bf16 evaluation forward: SATRANS_BF16_STACK=0 launches the layers one by one (satrans_layer_fwd_bf16)
        self.bf16_stack = os.environ.get("SATRANS_BF16_STACK", "1") != "0"
        # General path (configs[4]): SATRANS_GEN_SORTED=0 restores the order change at both ends of every layer
        self.sorted_acts = os.environ.get("SATRANS_GEN_SORTED", "1") != "0"
        self.rolling_flush = {"1": "tail", "0": ""}.get(os.environ.get("SATRANS_ROLLING_FLUSH", "0"),
                                                        os.environ.get("SATRANS_ROLLING_FLUSH", "0"))
        self._roll = None
        self._roll_done = None
        # Several ranks: "owner" (default) = every rank owns a contiguous 1/N slice of the large tables' rows and is the only one
        # to step them (per-rank optimizer work and traffic independent of N); "replicated" = round 2's exchange, every rank
        # applies every rank's updates (satrans_amd/parallel.py).
        self.dp_mode = os.environ.get("SATRANS_DP_MODE", "owner")
        self._x_src = None               # (row buffer, row index per (b, f)): layer 0 reads its tokens from here instead of the arena
        self._owner_world = 0            # > 0: large-table rows outside this rank's slice may be stale until _sync_replicas()
        self._replicas_stale = False
        self.reg_small = torch.zeros(1, dtype=torch.float64, device=self.dev)
        # The step's structure.  These are attributes, not environment switches: the shipped values are the ones below; the
        # parity tests flip them to hold the fused forms against the plain ones bit for bit.
        # the first layer reads its tokens straight from the embedding arena (no [B,F,D] gather output)
        self.fuse_gather = True
        # the last layer of a training step as one launch with the head fused in (satrans_layer_bwd_head): no forward launch
        # for that layer, no head launches, no [B,F,D] round trip of its output and gradient
        self.fuse_head = True
        # train_step(next_X=...): the next batch's ids -> rows, sort and bucketing on a side stream under this step's tail
        self.prefetch = True
        # owner form, several ranks: where the id exchange of a batch runs.  Default (round 6): inside the step, on the default
        # process group - one communicator, one issue order, nothing to deadlock, and on the one setting this
        # tooling can measure (one rank through RCCL) also the fastest: 1.084 ms/step against 1.20 with round 5's exchange a
        # step ahead on a second communicator and 1.91 with that exchange ordered behind the step's gradient collectives by
        # events (profiles/bench_r06_owner_form_one_rank_rccl*.json).  SATRANS_OWNER_PREFETCH=1 (or engine.owner_prefetch =
        # True, the same on every rank) selects the ordered a-step-ahead form (_prepare_owner_async) - what an 8-GPU node
        # would have to arbitrate, which no box available to this repository can.
        self.owner_prefetch = os.environ.get("SATRANS_OWNER_PREFETCH", "0") == "1"
        # arithmetic of the embedding tables' Adam update (satrans_adam_hparams.arith): "fast" = hardware sqrt / reciprocal, every
        # update within 3.2e-7 relative of torch's; "exact" = torch.optim.Adam's fp32 operations bit for bit
        # (SATRANS_ADAM_ARITH=exact).  Either way the streaming, gathered-row and lazy forms leave identical bits.
        self.adam_arith = os.environ.get("SATRANS_ADAM_ARITH", "fast")
        if self.adam_arith not in ("fast", "exact"):
            raise ValueError(f"SATRANS_ADAM_ARITH={self.adam_arith!r}: 'fast' or 'exact'")
        # phases that `phase()` does not bracket with recorded events even while `timers` is set (bench.py: the fused layer kernels,
        # whose own durations come from satrans_kernel_timing without a marker in the queue)
        self.untimed_phases = frozenset()
        # ... forked in FRONT of the last backward kernel, on a stream of the lowest priority: its workgroups find no room beside
        # that kernel's and start as its CUs come free, without the ~14 us a fork behind the kernel costs
        self.prep_early = True
        # one reduction launch for all layers' weight-gradient slabs (satrans_layer_bwd_reduce)
        self.defer_reduce = True
        # ... on a stream of its own beside the touched-row kernels, with the scenario-table backward
        self.side_tail = True
        # a PIPELINED step (train_step(next_X=...)) clears the next step's gradient buffer behind its own flat Adam launch;
        # param.grad is then not readable after the step (documented in train_step).  A step without `next_X` never does this:
        # its gradients stay in place until the next step, as the reference's do until the next zero_grad()
        self.preclear = True
        self._side_tail = None
        self._tail_done = None
        self._prep = None
        # The forward of a training step leaves the attention's softmax numerators / statistics / output for its backward, which
        # copies them into LDS (global_load_lds) instead of recomputing them - where the fused kernels are built for it (the
        # (32, 64, 4) MetaNet shape with one shared table; +72 MB per layer at B = 8192).  With fp32 products (round 4): forward
        # +7 us, backward -23 us per layer (A/B on one box: -13 to -37 us per step); the last layer has no forward launch and
        # keeps recomputing.  (Attribute; False = every backward recomputes: tests.)
        self.save_attention = True
        # evaluation forwards (predict / evaluate / model.eval()(X)) with the dense products in bf16 on the matrix pipe
        # (csrc/layer_fwd_bf16.hip; BASELINE.json configs[1]).  Off by default: fp32 is the parity path.  Also
        # model.set_forward_precision("bf16" | "fp32").
        self.fwd_bf16 = os.environ.get("SATRANS_FWD_BF16", "0") == "1"
        # SATRANS_SPLIT_TABLES=1: use the small/large table classes of the multi-rank step on a single rank too (tests)
        self.force_split = os.environ.get("SATRANS_SPLIT_TABLES", "0") == "1"
        self.last_step = None            # [R] int32: last Adam step applied to every table row
        self._hp_table = None            # [cap, 2] fp64: (fp32(lr / (1 - beta1^s)), 1 / fp32(sqrt(1 - beta2^s))) for step s
        self._hp_cfg = None
        self._lazy_pending = False

    # ------------------------------------------------------------------------------------------------
    def _stream(self):
        return N.stream_handle(self.dev)

    class _Phase:
        def __init__(self, eng, name):
            self.eng, self.name = eng, name

        def __enter__(self):
            self.on = self.eng.timers is not None and self.name not in self.eng.untimed_phases
            if self.on:
                self.t0 = torch.cuda.Event(enable_timing=True)
                self.t0.record(torch.cuda.current_stream(self.eng.dev))

        def __exit__(self, *exc):
            if self.on:
                t1 = torch.cuda.Event(enable_timing=True)
                t1.record(torch.cuda.current_stream(self.eng.dev))
                self.eng.timers.setdefault(self.name, []).append((self.t0, t1))

    def phase(self, name):
        return PathEngine._Phase(self, name)

    def phase_ms(self) -> Dict[str, float]:
        """MEDIAN milliseconds per occurrence of every timed phase (synchronises).  The median, not the mean: a handful of samples
        per phase, and one of them regularly catches a hiccup (a first-touch page fault, an allocator call) that says nothing
        about the phase."""
        torch.cuda.synchronize(self.dev)
        out = {}
        for k, v in (self.timers or {}).items():
            ts = sorted(a.elapsed_time(b) for a, b in v)
            out[k] = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
        return out

    def workspace(self, B: int) -> dict:
        ws = self._ws.get(B)
        if ws is not None:
            return ws
        dev, F, D = self.dev, self.F, self.D
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        ws = dict(
            sid=torch.empty(B, **i32), order=torch.empty(B, **i32), seg=torch.empty(self.S + 1, **i32),
            bucket=torch.empty(int(self.lib.satrans_bucket_workspace_bytes(B, self.S)), dtype=torch.uint8, device=dev),
            acts=[torch.empty(B, F, D, **f32) for _ in range(self.L + 1)],
            rows=torch.empty(B, F, **i32),
            prob=torch.empty(B, **f32), logit=torch.empty(B, **f32),
        )
        self._ws[B] = ws
        # Which layer kernels serve this shape: the fused / LDS-resident ones behind satrans_layer_fwd/_bwd, or - shapes
        # they cannot hold, e.g. 64 fields x embedding_dim 64 x MetaNet hidden 128 - the general path (csrc/layer_generic.hip),
        # which keeps its activations in HBM: one `saved` buffer per layer between forward and backward, one shared scratch.
        # SATRANS_GENERIC=1 forces the general path where it is supported (tests).
        probe = self._layer_desc(ws, 0, B, None, None, False)
        fits = int(self.lib.satrans_layer_bwd_slab_floats(C.byref(probe))) >= 0
        can = bool(self.lib.satrans_layer_generic_supported(C.byref(probe)))
        # (`gate` / `bilinear`: fused kernels at the shapes they are built for - D = 32 / 4 heads, D = 16 / 2 heads - since round 3;
        # elsewhere the general path rather than the LDS kernels, whose backward is an order of magnitude slower)
        choice = os.environ.get("SATRANS_GENERIC")
        fused = fits and bool(self.lib.satrans_layer_fused_supported(C.byref(probe)))
        ws["generic"] = can and (choice == "1" or not fits or (choice is None and (self.gate or self.bilinear) and not fused))
        if ws["generic"]:
            n = int(self.lib.satrans_layer_generic_saved_floats(C.byref(probe)))
            ws["gen_saved"] = [torch.empty(n, **f32)]
        return ws

    def train_workspace(self, B: int, world: int = 1, exchange: bool = False) -> dict:
        ws = self.workspace(B)
        key = ("train", world, exchange)
        if key in ws:
            return ws
        dev, F, D, lib = self.dev, self.F, self.D, self.lib
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        f64 = dict(dtype=torch.float64, device=dev)
        n_loc = B * F
        n_s = B * self.F_small                       # sorted positions [0, n_s): small tables; [n_s, n_loc): large tables
        n_big = (n_loc - n_s) * world                # large-table positions of ALL ranks
        n_max = max(n_loc, n_big)
        if "dact" not in ws:                                    # independent of the rank count: allocated once
            ws["dact"] = [torch.empty(B, F, D, **f32) for _ in range(2)]
            desc = self._layer_desc(ws, 0, B, None, None, True)
            # partial rows of the head's weight gradient: one per 8 samples (head_kernel) or one per workgroup of the fused
            # last-layer step (satrans_layer_bwd_head), whichever is more
            n_head = max(int(lib.satrans_head_scratch_floats(B, F * D, self.n_dense)),
                         int(lib.satrans_layer_bwd_head_scratch_floats(C.byref(desc), self.n_dense)))
            ws["head_scratch"] = torch.empty(n_head, **f32)
            if ws["generic"]:
                n = int(lib.satrans_layer_generic_saved_floats(C.byref(desc)))
                while len(ws["gen_saved"]) < self.L:            # one per layer
                    ws["gen_saved"].append(torch.empty(n, **f32))
                ws["gen_scratch"] = torch.empty(int(lib.satrans_layer_generic_scratch_floats(C.byref(desc))), **f32)
                ws["slabs"] = torch.empty(1, **f32)
            else:
                ws["slabs"] = torch.empty(int(lib.satrans_layer_bwd_slab_floats(C.byref(desc))), **f32)
                # what the forward leaves for the backward of the same step (softmax numerators, 1 / sum, keep words, attention
                # output): the backward then skips its attention-forward phase.  One buffer per layer (72 MB at B = 8192).
                ws["attn_save_floats"] = int(lib.satrans_layer_attn_save_floats(C.byref(desc)))
                ws["attn_save"] = [None] * self.L
            ws["sorted_rows"] = torch.empty(n_loc, **i32)       # this rank's rows, sorted, and their source positions
            ws["src"] = torch.empty(n_loc, **i32)
            ws["touched"] = torch.empty((self.total_rows + 31) // 32, **i32)
            ws["reg_unused"] = torch.zeros(int(lib.satrans_embed_reg_partials(self.total_rows, max(n_s, 1), D)), **f64)
            ws["reg_rows"] = torch.zeros(max(1, int(lib.satrans_embed_adam_rows_partials(max(self.small_rows, 1), D))), **f64)
        if exchange and ws.get("_exch_n_big", -1) != n_big:
            ws["g_sorted"] = torch.empty(max(n_big, 1), **i32)  # every rank's large-table rows, sorted
            ws["g_src"] = torch.empty(max(n_big, 1), **i32)
            ws["packed"] = torch.empty(max(n_loc - n_s, 1), D, **f32)
            ws["replay_reg_g"] = torch.zeros((max(n_big, 1) * D + 255) // 256, **f64)
            ws["_exch_n_big"] = n_big
        if ws.get("_train_n_max", -1) != n_max:                 # sized by the longest (row, position) list of the step
            ws["iota"] = torch.arange(n_max, **i32)             # positions for the (row, position) sorts, written once
            ws["sort_ws"] = torch.empty(int(lib.satrans_embed_sort_workspace_bytes(n_max, self.total_rows)),
                                        dtype=torch.uint8, device=dev)
            ws["partial_ws"] = torch.empty(int(lib.satrans_embed_partial_ws_floats(n_max, D)), **f32)
            # [untouched | touched-row kernels | replay of this rank's rows]: one fixed-order sum per step covers all three
            n_reg = int(lib.satrans_embed_reg_partials(self.total_rows, n_max, D))
            ws["reg_partials"] = torch.zeros(n_reg + (n_loc * D + 255) // 256, **f64)
            ws["replay_reg"] = ws["reg_partials"][n_reg:]
            ws["_train_n_max"] = n_max
        ws[key] = True
        return ws

    # ------------------------------------------------------------------------------------------------
    # scenario tables: row s = encoder(relu(scenario_embedding[s]))  (reference satrans.py:203-234, evaluated
    # once per scenario instead of once per sample - SURVEY.md §0)
    # ------------------------------------------------------------------------------------------------
    def scenario_tables(self, grad: bool) -> torch.Tensor:
        """-> [L, 2, S, P'] when 'pos' is in the flag (role 0 = Q, 1 = K), else [1, 1, S, P']."""
        m = self.m
        st = self._stream()
        if self.onlyemb:                                                            # satrans.py:173-176: relu(embedding of width P)
            emb = m.domain_embeddings.weight
            tab = torch.empty(1, 1, self.S, self.P, dtype=torch.float32, device=self.dev)
            N.check(self.lib.satrans_scenario_relu_fwd(emb.data_ptr(), emb.numel(), tab.data_ptr(), st),
                    "satrans_scenario_relu_fwd")
            return tab
        lin = m.domain_map_dnn_Q.linears[0]
        if self.plain_tabs:
            # common case (one scenario column, no 'pos'): one kernel; its backward is one call in backward()
            emb, rows, De = m.domain_embeddings.weight, self.S, m.domain_embeddings.weight.shape[1]
            shape = (1, 1, self.S, self.P)
        else:
            # several scenario columns (satrans.py:205-207: mean of the columns' rows per id tuple) and / or 'pos'
            # (:225-234: [scenario | layer id + q/k id] rows, one table per (layer, role)): assemble the encoder's input rows
            lay = m.layerid_embeddings.weight.data_ptr() if self.pos else None
            role = m.qkvid_embeddings.weight.data_ptr() if self.pos else None
            D0 = m.domain_embeddings.weight.shape[1] if not self.multi else self._scen_tables[0].weight.shape[1]
            N.check(self.lib.satrans_scenario_inputs_fwd(
                self._scen_ptrs, self._scen_index.data_ptr() if self._scen_index is not None else None,
                len(self._scen_tables), self.S, D0, lay, role, self.L, self._E.data_ptr(), st), "satrans_scenario_inputs_fwd")
            emb, rows, De = self._E, self.LR * self.S, self.De
            shape = (self.L, 2, self.S, self.P) if self.pos else (1, 1, self.S, self.P)
        tab = torch.empty(shape, dtype=torch.float32, device=self.dev)
        N.check(self.lib.satrans_scenario_table_fwd(emb.data_ptr(), lin.weight.data_ptr(), lin.bias.data_ptr(), rows, De, self.P,
                                                    tab.data_ptr(), st), "satrans_scenario_table_fwd")
        return tab

    def scenario_tables_backward(self, g_tabs: torch.Tensor) -> None:
        """Gradient of the generated-weight tables -> scenario / position embeddings and the encoder (flat gradient views)."""
        m, st = self.m, self._stream()
        if self.onlyemb:
            emb = m.domain_embeddings.weight
            N.check(self.lib.satrans_scenario_relu_bwd(emb.data_ptr(), g_tabs.data_ptr(), emb.numel(),
                                                       self._grad_view("domain_embeddings.weight").data_ptr(), st),
                    "satrans_scenario_relu_bwd")
            return
        lin = m.domain_map_dnn_Q.linears[0]
        gW = self._grad_view("domain_map_dnn_Q.linears.0.weight").data_ptr()
        gb = self._grad_view("domain_map_dnn_Q.linears.0.bias").data_ptr()
        if self.plain_tabs:
            emb = m.domain_embeddings.weight
            N.check(self.lib.satrans_scenario_table_bwd(
                emb.data_ptr(), lin.weight.data_ptr(), g_tabs.data_ptr(), self.S, emb.shape[1], self.P,
                self._grad_view("domain_embeddings.weight").data_ptr(), gW, gb, self._tab_ws.data_ptr(), st),
                    "satrans_scenario_table_bwd")
            return
        self._gE.zero_()
        N.check(self.lib.satrans_scenario_table_bwd(
            self._E.data_ptr(), lin.weight.data_ptr(), g_tabs.data_ptr(), self.LR * self.S, self.De, self.P,
            self._gE.data_ptr(), gW, gb, self._tab_ws.data_ptr(), st), "satrans_scenario_table_bwd")
        keys = [f"domain_embedding_dict.{c.embedding_name}.weight" for c in m.domain_feature_columns] if self.multi \
            else ["domain_embeddings.weight"]
        g_ptrs = (C.c_void_p * len(keys))(*[self._grad_view(k).data_ptr() for k in keys])
        D0 = self._scen_tables[0].weight.shape[1]
        N.check(self.lib.satrans_scenario_inputs_bwd(
            g_ptrs, self._scen_rows, self._scen_index.data_ptr() if self._scen_index is not None else None,
            len(keys), self.S, D0, self._gE.data_ptr(),
            self._grad_view("layerid_embeddings.weight").data_ptr() if self.pos else None,
            self._grad_view("qkvid_embeddings.weight").data_ptr() if self.pos else None, self.L, st),
                "satrans_scenario_inputs_bwd")

    def _layer_desc(self, ws, l, B, x, tabs, training, fuse=False, attn_save=False, sorted_io=False) -> N.LayerDesc:
        m = self.m
        lay = m.domain_int_layers[l]
        d = N.LayerDesc()
        d.B, d.F, d.D, d.H, d.U, d.S = B, self.F, self.D, self.H, self.U, self.S
        d.flags = self.flags | (N.TRAIN if training else 0)
        if sorted_io:      # general path, training step: interior activations (and their gradients) stay scenario-sorted
            d.flags |= (N.X_SORTED if l > 0 else 0) | (N.Y_SORTED if l < self.L - 1 else 0)
        d.layer = l
        d.drop_p = self.drop_p
        d.seed, d.step = self.drop_seed, self.drop_step & 0xFFFFFFFF
        d.x = N.ptr(x) if x is not None else ws["acts"][l].data_ptr()
        d.x_rows = None
        d.attn_save = None
        if attn_save and self.save_attention and ws.get("attn_save_floats", 0) > 0:
            # one buffer per layer that HAS a forward launch in a training step, allocated when that layer first asks (with the
            # head fused in the last layer never does: 113 MB less at B = 8192)
            if ws["attn_save"][l] is None:
                ws["attn_save"][l] = torch.empty(ws["attn_save_floats"], dtype=torch.float32, device=self.dev)
            d.attn_save = ws["attn_save"][l].data_ptr()
        if fuse and l == 0:
            # gather fused into the first layer: token (b, f) is read straight from the embedding arena through the row
            # numbers the rows-only gather launch left in ws["rows"]; [B,F,D] is never written nor read back.
            # (An out-of-range id - an IndexError in the reference - is recorded as the first row of the field's table and
            # flagged in `status`; the fused layer reads that row where the standalone gather writes zeros.  Either way the
            # batch is invalid and raise_if_bad_ids() raises before any result is handed to the caller.)
            d.x, d.x_rows = self.m.embedding_arena.data_ptr(), ws["rows"].data_ptr()
            if self._x_src is not None:      # owner-mode step: the rows came from their owners (this step's current values)
                d.x, d.x_rows = self._x_src[0].data_ptr(), self._x_src[1].data_ptr()
        d.sid, d.order, d.seg = ws["sid"].data_ptr(), ws["order"].data_ptr(), ws["seg"].data_ptr()
        d.w_query, d.w_key, d.w_value = lay.W_Query.data_ptr(), lay.W_Key.data_ptr(), lay.W_Value.data_ptr()
        d.w_out = lay.Out_linear.weight.data_ptr()
        d.ln_g, d.ln_b = lay.layer_norm.weight.data_ptr(), lay.layer_norm.bias.data_ptr()
        d.lnq_g = lay.Q_meta_mlp.ffn_layer_norm.weight.data_ptr()
        d.lnq_b = lay.Q_meta_mlp.ffn_layer_norm.bias.data_ptr()
        d.lnk_g = lay.K_meta_mlp.ffn_layer_norm.weight.data_ptr()
        d.lnk_b = lay.K_meta_mlp.ffn_layer_norm.bias.data_ptr()
        if tabs is not None:
            tq, tk = self._layer_tables(tabs, l)
            d.tab_q, d.tab_k = tq.data_ptr(), tk.data_ptr()
            d.tab_stride = tabs.shape[-1]
        else:                                                # size queries only
            d.tab_q = d.tab_k = ws["sid"].data_ptr()
            d.tab_stride = 2 * self.D * self.U
        return d

    def bwd_kernel_name(self) -> str:
        """Name of the fused backward kernel this process launches (profiles and bench.py's roofline line match on it)."""
        return "layer_bwd_fused_kernel"

    def _layer_tables(self, tabs, l):
        if self.pos:
            return tabs[l, 0], tabs[l, 1]
        return tabs[0, 0], tabs[0, 0]

    # ------------------------------------------------------------------------------------------------
    def _prepare_input(self, X):
        from .inputs import PackedInput
        self._dense_override = None
        if isinstance(X, PackedInput):            # integer ids + a separate float block for the dense features
            if self.n_dense:
                dn = X.dense
                if dn.dtype != torch.float32 or not dn.is_contiguous():
                    dn = dn.float().contiguous()
                if dn.shape[1] != self.n_dense:
                    raise ValueError(f"expected {self.n_dense} dense columns, got {dn.shape[1]}")
                self._dense_override = dn
            X = X.ids
        N.require_gpu(X, "SATrans.forward input")
        if X.dim() != 2 or X.shape[1] < self.n_cols:
            raise ValueError(f"expected X of shape [B, {self.n_cols}], got {tuple(X.shape)}")
        if X.dtype not in (torch.float32, torch.int32, torch.int64):
            X = X.float()
        if not X.is_contiguous():
            X = X.contiguous()
        if X.dtype != torch.float32 and self.n_dense and self._dense_override is None:
            raise NotImplementedError("integer id matrix together with dense features: pass inputs.PackedInput(ids, dense)")
        return X

    def _bucket(self, X, ws):
        lib, B, st = self.lib, X.shape[0], self._stream()
        sx, sidt, sstride, scol = X, N.id_dtype_of(X), X.stride(0), self.dom_col
        if self.multi:   # composite scenario id of every sample (a few elementwise torch ops on [B, k] ids)
            sx = (X.index_select(1, self._multi_cols).long() * self._multi_strides).sum(1).to(torch.int32).contiguous()
            sidt, sstride, scol = N.ID_I32, 1, 0
        N.check(lib.satrans_bucket_scenarios(sx.data_ptr(), sidt, sstride, scol, B, self.S,
                                             ws["sid"].data_ptr(), ws["order"].data_ptr(), ws["seg"].data_ptr(),
                                             self.status.data_ptr(), ws["bucket"].data_ptr(), ws["bucket"].numel(), st),
                "satrans_bucket_scenarios")

    def _run_forward(self, X, ws, training, tabs, att_list=None, rows_ready=False, save_attn=False, n_layers=None,
                     bucket_ready=False, sorted_io=False, eval_head=False) -> bool:
        """-> True when the head ran too (`eval_head`: the bf16 evaluation stack with the head behind it in the same launch)."""
        lib, B, st = self.lib, X.shape[0], self._stream()
        idt = N.id_dtype_of(X)
        if not bucket_ready:
            self._bucket(X, ws)
        fuse = self.fuse_gather or self._x_src is not None
        ws["acts0_of"] = None if fuse else X
        ws["acts_sorted"] = bool(sorted_io)
        ws["acts_stacked"] = False
        if not (fuse and rows_ready):
            with self.phase("gather_fwd"):
                N.check(lib.satrans_gather_fwd(self.m.embedding_arena.data_ptr(), self.row_span.data_ptr(),
                                               self.cols.data_ptr(), X.data_ptr(), idt, X.stride(0), B, self.F, self.D,
                                               None if fuse else ws["acts"][0].data_ptr(), ws["rows"].data_ptr(),
                                               self.status.data_ptr(), st), "satrans_gather_fwd")
        self._last_X = X
        self._stepped_since_forward = False
        # evaluation on the bf16 pipe: the whole stack as ONE launch where it is built (a tile's rows stay in LDS between the
        # layers); `layer_outputs()` then only has the last layer's - callers that want every layer's set engine.bf16_stack = False
        if self.fwd_bf16 and self.bf16_stack and not training and att_list is None and n_layers is None and not ws["generic"] \
                and 1 < self.L <= 4:
            descs = [self._layer_desc(ws, l, B, None, tabs, False, fuse) for l in range(self.L)]
            arr = (C.POINTER(N.LayerDesc) * self.L)(*[C.pointer(d_) for d_ in descs])
            if lib.satrans_stack_fwd_bf16_supported(self.L, arr):
                ws["acts_stacked"] = True       # (the layers' outputs never left LDS: layer_outputs() re-runs them one by one)
                with self.phase("layer_fwd"):
                    if eval_head and self.n_dense <= 2:
                        self._join_prob_readers()
                        hd = self._head_desc(X, ws, None)
                        N.check(lib.satrans_stack_fwd_bf16_head(self.L, arr, C.byref(hd), st), "satrans_stack_fwd_bf16_head")
                        return True
                    N.check(lib.satrans_stack_fwd_bf16(self.L, arr, ws["acts"][self.L].data_ptr(), st), "satrans_stack_fwd_bf16")
                return False
        for l in range(self.L if n_layers is None else n_layers):
            desc = self._layer_desc(ws, l, B, None, tabs, training, fuse, attn_save=save_attn, sorted_io=sorted_io)
            att = att_list[l].data_ptr() if att_list is not None else None
            # (layer 0 reads its tokens straight from the embedding arena - the gather fused in: its own phase name, so that what
            #  the random row reads cost shows next to the other layers' time)
            with self.phase("layer_fwd_gather" if (fuse and l == 0 and training) else "layer_fwd"):
                if ws["generic"] and not (self.fwd_bf16 and not training and att is None
                                          and lib.satrans_layer_fwd_bf16_supported(C.byref(desc))):
                    saved = ws["gen_saved"][l if len(ws["gen_saved"]) > l else 0]
                    N.check(lib.satrans_layer_fwd_generic(C.byref(desc), ws["acts"][l + 1].data_ptr(), att, saved.data_ptr(), st),
                            "satrans_layer_fwd_generic")
                elif self.fwd_bf16 and not training and att is None and lib.satrans_layer_fwd_bf16_supported(C.byref(desc)):
                    N.check(lib.satrans_layer_fwd_bf16(C.byref(desc), ws["acts"][l + 1].data_ptr(), st), "satrans_layer_fwd_bf16")
                else:
                    N.check(lib.satrans_layer_fwd(C.byref(desc), ws["acts"][l + 1].data_ptr(), att, st), "satrans_layer_fwd")

    def after_step(self, fn, *tensors) -> None:
        """Launches that only READ what the step just produced - the probabilities of `last_prob()`, the labels: `fit`'s per-step
        metrics - go on a stream of their own that starts behind the step's last backward kernel, beside the tail kernels
        (reduction, scenario-table backward, flat Adam) and the touched-row chain, instead of between two steps of the launch
        stream.  `fn()` is called with that stream current;
        `tensors` are the caller's operands, kept from being recycled while that stream uses them.  The next launch that
        overwrites the probabilities waits for it (`_join_prob_readers`).  Without such a stream (several ranks, the fused head
        switched off) `fn()` simply runs on the launch stream."""
        fork = getattr(self, "_last_fork", None)
        if self._side_tail is None or getattr(self, "_flat_done", None) is None or not self.side_tail or fork is None:
            fn()
            return
        # A stream of its own that starts where the tail stream starts - right behind the last backward kernel - so that the
        # launch runs in the seam between two steps, where the CUs are idle (the touched-row chain, the reduction and the
        # scenario-table backward are small).  Queued BEHIND the tail stream's kernels (round 5) the one-workgroup metrics kernel
        # of `fit` (57 us, ~75 KB of LDS: no room beside a layer kernel's workgroup) was still running when the next step's first
        # layer kernel started: one of that persistent kernel's 256 workgroups waited for its CU and the whole launch ended
        # ~44 us late - fit(verbose=1) ran at 0.9975 ms per step against 0.9535 with verbose=0 (profiles/r06_fit_epoch.txt).
        st = _shared_stream(self.dev, "reader", lambda: torch.cuda.Stream(self.dev))
        st.wait_event(fork)
        with torch.cuda.stream(st):
            fn()
            self._prob_read = torch.cuda.Event()
            self._prob_read.record(st)
        for t in tensors:
            t.record_stream(st)

    def _join_prob_readers(self) -> None:
        ev, self._prob_read = getattr(self, "_prob_read", None), None
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)

    def _head(self, X, ws, y=None, train_ws=None):
        lib, B, st = self.lib, X.shape[0], self._stream()
        m = self.m
        self._join_prob_readers()
        dense_ptr = X.data_ptr() if self.n_dense else None
        dcols = self.dense_cols.data_ptr() if self.n_dense else None
        dstride = X.stride(0)
        if self.n_dense and getattr(self, "_dense_override", None) is not None:
            if getattr(self, "_dense_iota", None) is None:
                self._dense_iota = torch.arange(self.n_dense, dtype=torch.int32, device=self.dev)
            dense_ptr, dcols, dstride = self._dense_override.data_ptr(), self._dense_iota.data_ptr(), self._dense_override.stride(0)
        if y is None:
            N.check(lib.satrans_head(ws["acts"][self.L].data_ptr(), dense_ptr, dstride, dcols, self.n_dense, B,
                                     self.F * self.D, m.dnn_linear.weight.data_ptr(), m.dnn_linear.bias.data_ptr(),
                                     ws["prob"].data_ptr(), ws["logit"].data_ptr(), None, None, None, None, None, None,
                                     st), "satrans_head")
        else:
            gw, gb = self._grad_view("dnn_linear.weight"), self._grad_view("dnn_linear.bias")
            kind = {"binary_crossentropy": 0, "mse": 1, "mae": 2}[getattr(m, "loss_func", "binary_crossentropy")]
            N.check(lib.satrans_head_loss(ws["acts"][self.L].data_ptr(), dense_ptr, dstride, dcols, self.n_dense, B,
                                          self.F * self.D, m.dnn_linear.weight.data_ptr(), m.dnn_linear.bias.data_ptr(),
                                          ws["prob"].data_ptr(), ws["logit"].data_ptr(), y.data_ptr(),
                                          self.loss_sum.data_ptr(), ws["dact"][0].data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                          ws["head_scratch"].data_ptr(), kind, st), "satrans_head_loss")

    def _head_desc(self, X, ws, y) -> N.HeadDesc:
        """Head operands of satrans_layer_bwd_head (the last layer of a training step with the head fused in)."""
        m = self.m
        h = N.HeadDesc()
        h.w, h.bias = m.dnn_linear.weight.data_ptr(), m.dnn_linear.bias.data_ptr()
        h.labels = y.data_ptr() if y is not None else None
        h.n_dense = self.n_dense
        h.dense, h.dense_stride, h.h_dense_cols = None, 0, None
        if self.n_dense:
            if getattr(self, "_dense_override", None) is not None:
                h.dense, h.dense_stride = self._dense_override.data_ptr(), self._dense_override.stride(0)
                cols = list(range(self.n_dense))
            else:
                h.dense, h.dense_stride = X.data_ptr(), X.stride(0)
                cols = self._dense_cols_host
            self._head_cols_keep = (C.c_int32 * len(cols))(*cols)          # (kept alive until the call returns)
            h.h_dense_cols = self._head_cols_keep
        h.loss_kind = {"binary_crossentropy": 0, "mse": 1, "mae": 2}[getattr(m, "loss_func", "binary_crossentropy")]
        h.prob, h.logit = ws["prob"].data_ptr(), ws["logit"].data_ptr()
        h.loss_sum = self.loss_sum.data_ptr()
        if self.flat_g is not None:
            h.g_w, h.g_b = self._grad_view("dnn_linear.weight").data_ptr(), self._grad_view("dnn_linear.bias").data_ptr()
        h.scratch = ws["head_scratch"].data_ptr() if "head_scratch" in ws else None
        return h

    def _fuse_head(self, X, ws, B) -> bool:
        """Whether this step runs its last layer as ONE launch with the head fused in (satrans_layer_bwd_head): the fused
        kernels, at most two dense columns.  (`fuse_head` False: the three separate calls.)"""
        key = "fuse_head"
        if key not in ws:
            ok = self.fuse_head and not ws["generic"] and self.L >= 1
            if ok:
                desc = self._layer_desc(ws, self.L - 1, B, None, None, True)
                ok = bool(self.lib.satrans_layer_bwd_head_supported(C.byref(desc), C.byref(self._head_desc(X, ws, None))))
                if ok:
                    need = int(self.lib.satrans_layer_bwd_head_scratch_floats(C.byref(desc), self.n_dense))
                    ok = 0 <= need <= ws["head_scratch"].numel()
            ws[key] = ok
        return ws[key]

    def forward(self, X: torch.Tensor, training: bool = False, capture_attention: bool = False) -> torch.Tensor:
        self._join_flat()
        self.flush_lazy()
        X = self._prepare_input(X)
        B = X.shape[0]
        ws = self.workspace(B)
        tabs = self.scenario_tables(grad=False)
        att_list = None
        if capture_attention:
            att_list = [torch.empty(self.H, B, self.F, self.F, dtype=torch.float32, device=self.dev)
                        for _ in range(self.L)]
        if training:
            self.drop_step += 1
        if not self._run_forward(X, ws, training, tabs, att_list, eval_head=True):
            self._head(X, ws)
        if att_list is not None:
            for layer, att in zip(self.m.domain_int_layers, att_list):
                layer.normalized_att_scores = att
        self._last_logit = ws["logit"]
        return ws["prob"].clone().unsqueeze(1)

    def last_prob(self) -> torch.Tensor:
        return self._last_prob

    def last_logit(self) -> torch.Tensor:
        return self._last_logit.clone().unsqueeze(1)

    def layer_outputs(self, B: int) -> List[torch.Tensor]:
        """[att_input, layer 0 output, ...] of the most recent forward at this batch size (tests, attention dumps).  With the
        gather fused into the first layer `att_input` was never materialised: the standalone gather kernel writes it now."""
        ws = self._ws[B]
        if ws.get("acts_stacked"):
            # the last forward ran the whole stack as one launch (bf16 evaluation): the same layers again, one launch each - the same
            # bits (tests) - so that every layer's output exists
            keep, self.bf16_stack = self.bf16_stack, False
            try:
                self._run_forward(self._last_X, ws, False, self.scenario_tables(grad=False))
            finally:
                self.bf16_stack = keep
        if ws.get("acts0_of") is None:
            if getattr(self, "_stepped_since_forward", False):
                # the gather was fused into layer 0 and the optimizer has since moved the rows it read: a re-gather would
                # NOT be the input that forward saw
                raise RuntimeError("layer_outputs() after train_step(): att_input was never materialised (gather fused into "
                                   "the first layer) and the tables have been updated since; call forward() / "
                                   "loss_and_grads() first, or set engine.fuse_gather = False")
            X = self._last_X
            N.check(self.lib.satrans_gather_fwd(self.m.embedding_arena.data_ptr(), self.row_span.data_ptr(),
                                                self.cols.data_ptr(), X.data_ptr(), N.id_dtype_of(X), X.stride(0), B, self.F,
                                                self.D, ws["acts"][0].data_ptr(), ws["rows"].data_ptr(),
                                                self.status.data_ptr(), self._stream()), "satrans_gather_fwd")
            ws["acts0_of"] = X
        acts = [a.clone() for a in ws["acts"]]
        if ws.get("acts_sorted"):
            # a training step of the general path left the interior activations in scenario-sorted order (sorted position p holds
            # sample order[p]): hand them out in the caller's order like every other
            order = ws["order"].long()
            for l in range(1, self.L):
                back = torch.empty_like(acts[l])
                back[order] = acts[l]
                acts[l] = back
        return acts

    def raise_if_bad_ids(self):
        if int(self.status.item()) != 0:
            self.status.zero_()
            raise IndexError("index out of range in self: an id (or scenario id) exceeds its vocabulary_size")

    # ------------------------------------------------------------------------------------------------
    # training
    # ------------------------------------------------------------------------------------------------
    def _opt_kind(self) -> str:
        return (getattr(self.m, "_adam_cfg", None) or {}).get("kind", "adam")

    def _ensure_train_state(self):
        m = self.m
        if self.flat_g is None:
            # [flat parameters | dense gradient of the small tables]: ONE all-reduce moves both
            n_flat = m.flat_params.numel()
            n_tabs = int(self.scenario_tables(grad=False).numel())
            # layout [dense gradient of the small tables | flat parameters | generated-weight tables]: what data-parallel ranks
            # all-reduce is the contiguous head, what ONE rank has to clear before a backward pass is the contiguous tail (the
            # small-table part is only written by the multi-rank / split-table step: 2.2 MB of the 2.9 MB at AliCCP size)
            n_small = self.small_rows * self.D
            self.flat_g_all = torch.zeros(n_small + n_flat + n_tabs, dtype=torch.float32, device=self.dev)
            self.g_small = self.flat_g_all[:n_small].view(self.small_rows, self.D)
            self.flat_g = self.flat_g_all[n_small:n_small + n_flat]               # gradient of the flat parameter buffer
            self.g_exchange = self.flat_g_all[:n_small + n_flat]
            self._g_tabs_flat = self.flat_g_all[n_small + n_flat:]                # gradient of the generated-weight table
            self._g_step_tail = self.flat_g_all[n_small:]
            # expose gradients the torch way: param.grad is a view into the flat gradient buffer
            for name, p in m._trainable_flat().items():
                off, cnt = m._flat_slices[name]
                p.grad = self.flat_g[off:off + cnt].view(p.shape)
        # optimizer state of the kind compile() chose: Adam's two moment arenas (+ the lazy form's per-row step), or ONE
        # accumulator arena for Adagrad (`sum`) / RMSprop (`square_avg`); plain SGD keeps none.  Allocated on first use of
        # that kind only: at configs[4] an arena is 25.6 GB.
        if self._opt_kind() == "adam":
            if self.adam_m is None:
                self.flat_m = torch.zeros_like(m.flat_params)
                self.flat_v = torch.zeros_like(m.flat_params)
                self.adam_m = torch.zeros_like(m.embedding_arena)
                self.adam_v = torch.zeros_like(m.embedding_arena)
                self.last_step = torch.zeros(self.total_rows, dtype=torch.int32, device=self.dev)
                self._flush_reg = torch.zeros(4096, dtype=torch.float64, device=self.dev)
        elif getattr(self, "_g_arena", None) is None:
            self._g_arena = torch.empty_like(m.embedding_arena)
            self._opt_state_arena = torch.zeros_like(m.embedding_arena)
            self._opt_state_flat = torch.zeros_like(m.flat_params)

    def _grad_view(self, name: str) -> torch.Tensor:
        off, cnt = self.m._flat_slices[name]
        return self.flat_g[off:off + cnt]

    def optimizer_state(self) -> dict:
        """Everything a new engine needs to continue this run (call flush_lazy() first: every row is then at adam_t).
        `kind` names the optimizer the state belongs to: "adam" carries the two moment arenas, "adagrad" / "rmsprop" their
        one accumulator (`acc_arena`, `acc_flat`), "sgd" only the step counters."""
        self._ensure_train_state()
        kind = self._opt_kind()
        st = dict(kind=kind, adam_t=self.adam_t, drop_step=self.drop_step)
        if kind == "adam":
            st.update(adam_m=self.adam_m, adam_v=self.adam_v, flat_m=self.flat_m, flat_v=self.flat_v)
        else:
            st.update(acc_arena=self._opt_state_arena, acc_flat=self._opt_state_flat)
        return st

    def load_optimizer_state(self, st: dict) -> None:
        kind = st.get("kind", "adam")
        if kind != self._opt_kind():
            raise ValueError(f"optimizer state of kind {kind!r} loaded into a model compiled with {self._opt_kind()!r}")
        self._ensure_train_state()
        self.adam_t = int(st["adam_t"])
        self.drop_step = int(st["drop_step"])
        if kind == "adam":
            self.adam_m.copy_(st["adam_m"].to(self.dev))
            self.adam_v.copy_(st["adam_v"].to(self.dev))
            self.flat_m.copy_(st["flat_m"].to(self.dev))
            self.flat_v.copy_(st["flat_v"].to(self.dev))
            self.last_step.fill_(self.adam_t)            # the state was taken after a flush: every row is current
            self._table_lr = self._table_dirty = self._lr_hist = self._lr_starts = None    # (rows of earlier steps are never replayed again)
            self._hp_table = None
        else:
            self._opt_state_arena.copy_(st["acc_arena"].to(self.dev))
            self._opt_state_flat.copy_(st["acc_flat"].to(self.dev))
        self._lazy_pending = False

    def reset_epoch_sums(self):
        self._join_flat()          # (a pipelined step's slab reduction adds to loss_sum on the tail stream)
        self._join_roll()
        self.loss_sum.zero_()
        self.reg_sum.zero_()
        self.reg_small.zero_()
        self.reg_roll.zero_()

    def epoch_sums(self):
        """(sum of this rank's sample losses, sum over the epoch's steps of the regulariser l2 * |tables|^2).  Collective in
        the owner form of several ranks: a rank accumulates the regulariser of the rows it owns, the total is their sum."""
        self.flush_lazy()          # the regulariser sums of postponed steps belong to this epoch
        reg = self.reg_sum + self.reg_roll       # (flush_lazy joined the rolling flush's stream)
        if self._owner_world:
            from . import parallel
            reg = parallel.all_reduce_scalars(self.reg_sum.clone()) + self.reg_small
        return float(self.loss_sum.item()), float(reg.item())

    def _hparams(self, l2: float, tables: bool = True) -> N.AdamHParams:
        """`tables`: the step of the embedding tables (every kernel form of it takes the engine's `adam_arith`); the flat
        dense parameters - 0.2 M floats, one short launch - always take torch's operations bit for bit."""
        cfg = self.m._adam_cfg
        b1, b2 = cfg["betas"]
        t = self.adam_t
        h = N.AdamHParams()
        h.lr_over_bc1 = cfg["lr"] / (1.0 - b1 ** t)
        h.bc2_sqrt = math.sqrt(1.0 - b2 ** t)
        h.beta1, h.beta2, h.eps, h.l2 = b1, b2, cfg["eps"], l2
        h.arith = N.ADAM_FAST if (tables and self.adam_arith == "fast") else N.ADAM_EXACT
        return h

    def _join_flat(self):
        """The launch stream waits for the flat Adam launch of the previous step when that ran on the reduction's stream."""
        ev, self._flat_done = getattr(self, "_flat_done", None), None
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)

    def _defer_reduce(self, ws, B) -> bool:
        if "defer" not in ws:
            ok = self.defer_reduce and not ws["generic"] and 1 <= self.L <= 8
            if ok:
                desc = self._layer_desc(ws, 0, B, None, None, True)
                ok = bool(self.lib.satrans_layer_bwd_deferred_supported(C.byref(desc)))
            if ok:
                ws["slabs_l"] = [ws["slabs"]] + [torch.empty_like(ws["slabs"]) for _ in range(self.L - 1)]
            ws["defer"] = ok
        return ws["defer"]

    def backward(self, X, y, ws, rows_ready=False, bucket_ready=False, after_layers=None, side_tail=False):
        """Forward (training mode as set by the caller) + loss + backward.  Leaves the dense gradients in
        `flat_g` and the gradient of the gathered rows in the returned tensor [B,F,D]."""
        lib, B, st = self.lib, X.shape[0], self._stream()
        m = self.m
        self._join_prepared()          # (a batch prepared on the side stream: its bucketing, behind the sorted rows the replay needed)
        # (one rank without table classes never writes the small tables' dense gradient: only the tail is cleared)
        from . import parallel as _par
        g_clear = self.flat_g_all if (_par.exchange_enabled() or self.force_split) else self._g_step_tail
        # With the reduction deferred to one launch on its own stream nothing writes the dense gradients before that launch: the
        # clear goes there too, off the launch stream (6 us at the head of every step)
        # (only with the fused head: the separate head launch adds dnn_linear's gradient in front of the reduction)
        clear_late = side_tail and self.defer_reduce and ws.get("defer") is True and ws.get("fuse_head") is True
        # ... or the previous step has done it already, behind its flat Adam launch on that stream and behind the event the next
        # forward waits for (`_precleared`; in front of that event it cost 12 us per step: the fill delays the event)
        pre, self._precleared = getattr(self, "_precleared", None), None
        pre_ev, pre_buf = pre if pre is not None else (None, None)
        precleared = pre_ev is not None and pre_buf is g_clear
        if not clear_late:
            self._join_flat()          # (the previous step's flat Adam may still be reading what is cleared here)
            if pre_ev is not None:
                torch.cuda.current_stream(self.dev).wait_event(pre_ev)   # (... and its clear must not land on what this step writes)
            g_clear.zero_()
        training = m.training
        if training:
            self.drop_step += 1
        modulated = bool(self.flags & (N.META_Q | N.META_K | N.BILINEAR))
        self._join_flat()
        tabs = self.scenario_tables(grad=modulated)                           # (HIP kernels: no autograd graph either way)
        g_tabs = self._g_tabs_flat.view(tabs.shape) if modulated else None      # zeroed with flat_g above
        fuse_head = self._fuse_head(X, ws, B)
        # general path: the layers of the stack hand their activations on in scenario-sorted order, and their gradients back
        # likewise (no order change at the ends of the interior layers: SATRANS_X_SORTED / SATRANS_Y_SORTED)
        sorted_io = bool(ws["generic"]) and self.sorted_acts and self.L > 1
        self._run_forward(X, ws, training, tabs.detach(), rows_ready=rows_ready, save_attn=True,
                          n_layers=self.L - 1 if fuse_head else None, bucket_ready=bucket_ready, sorted_io=sorted_io)
        if not fuse_head:
            with self.phase("head"):
                self._head(X, ws, y)
        # Fused kernels: the L backward kernels back to back, a slab buffer each, and ONE reduction launch for all of them (and the
        # fused head's partial rows) instead of one ~10 us reduction per layer between them (satrans_layer_bwd_reduce).
        defer = self._defer_reduce(ws, B)
        d_descs, d_slabs, d_grads, d_head = [], [], [], None
        cur = 0
        early_ev = None
        for l in reversed(range(self.L)):
            head_here = fuse_head and l == self.L - 1
            desc = self._layer_desc(ws, l, B, None, tabs.detach(), training, self.fuse_gather or self._x_src is not None,
                                    attn_save=not head_here, sorted_io=sorted_io)
            lay = f"domain_int_layers.{l}."
            gq = gk = glnq = glnk = None
            if modulated:
                tq, tk = self._layer_tables(g_tabs, l)
                gq, gk = tq.data_ptr(), tk.data_ptr()
            if self.metanet and self.flags & N.META_Q:
                glnq = self._grad_view(lay + "Q_meta_mlp.ffn_layer_norm.weight").data_ptr()
            if self.metanet and self.flags & N.META_K:
                kname = "K_meta_mlp" if m.domain_int_layers[l].K_meta_mlp is not m.domain_int_layers[l].Q_meta_mlp \
                    else "Q_meta_mlp"
                glnk = self._grad_view(lay + kname + ".ffn_layer_norm.weight").data_ptr()
            g_ptrs = (self._grad_view(lay + "W_Query").data_ptr(), self._grad_view(lay + "W_Key").data_ptr(),
                      self._grad_view(lay + "W_Value").data_ptr(), self._grad_view(lay + "Out_linear.weight").data_ptr(),
                      self._grad_view(lay + "layer_norm.weight").data_ptr(), glnq, glnk, gq, gk)
            if ws["generic"]:
                with self.phase("layer_bwd"):
                    N.check(lib.satrans_layer_bwd_generic(
                        C.byref(desc), ws["dact"][cur].data_ptr(), ws["dact"][1 - cur].data_ptr(), ws["gen_saved"][l].data_ptr(),
                        ws["gen_scratch"].data_ptr(), *g_ptrs, st), "satrans_layer_bwd_generic")
                cur = 1 - cur
                continue
            slabs = ws["slabs_l"][l] if defer else ws["slabs"]
            if defer:
                d_descs.append(desc)
                d_slabs.append(slabs.data_ptr())
                d_grads.append(g_ptrs)
            if head_here:
                # layer L-1 forward (recomputed) + head + loss + their backward: one launch, the layer's output never leaves the CU
                hdesc = self._head_desc(X, ws, y)
                self._join_prob_readers()
                with self.phase("layer_bwd_head"):
                    if defer:
                        d_head = hdesc
                        N.check(lib.satrans_layer_bwd_head_launch(C.byref(desc), C.byref(hdesc), ws["dact"][1 - cur].data_ptr(),
                                                                  slabs.data_ptr(), st), "satrans_layer_bwd_head_launch")
                    else:
                        N.check(lib.satrans_layer_bwd_head(C.byref(desc), C.byref(hdesc), ws["dact"][1 - cur].data_ptr(),
                                                           slabs.data_ptr(), *g_ptrs, st), "satrans_layer_bwd_head")
                cur = 1 - cur
                continue
            untimed_launch = False
            if l == 0 and self.L >= 2 and after_layers is not None and self.prep_early and defer and \
                    (self.timers is None or "layer_bwd" in self.untimed_phases):
                # the next batch's preprocessing forks in front of the LAST backward kernel, on a low-priority stream - its workgroups
                # find no room beside that kernel's and fill the CUs as they come free (`prep_early`).  No marker may sit between
                # the two backward kernels: an event recorded there, or the start event of a launch under satrans_kernel_timing,
                # holds the second one back for a few us, which is all the head start the sort needs to take 19 CUs first (measured:
                # that kernel 203 -> 240-255 us).  So: not on steps whose layer phases are bracketed by recorded events, and on
                # steps under satrans_kernel_timing THIS launch goes untimed (layer 1's launch of the same kernel is the sample).
                early_ev = torch.cuda.Event()
                early_ev.record(torch.cuda.current_stream(self.dev))
                untimed_launch = self.timers is not None and bool(lib.satrans_kernel_timing(0))
            with self.phase("layer_bwd"):
                if defer:
                    N.check(lib.satrans_layer_bwd_launch(C.byref(desc), ws["dact"][cur].data_ptr(), ws["dact"][1 - cur].data_ptr(),
                                                         slabs.data_ptr(), st), "satrans_layer_bwd_launch")
                else:
                    N.check(lib.satrans_layer_bwd(C.byref(desc), ws["dact"][cur].data_ptr(), ws["dact"][1 - cur].data_ptr(),
                                                  slabs.data_ptr(), *g_ptrs, st), "satrans_layer_bwd")
            if untimed_launch:
                lib.satrans_kernel_timing(1)
            if early_ev is not None:
                # (queued by the host BEHIND the kernel it must not overtake, waiting for the event recorded in front of it)
                after_layers(early_ev)
                after_layers, early_ev = None, None
            cur = 1 - cur

        def finish():
            # what only the dense-parameter step needs: the slab reduction and the scenario-table backward
            if d_descs:
                n = len(d_descs)
                descs_a = (C.POINTER(N.LayerDesc) * n)(*[C.pointer(d_) for d_ in d_descs])
                slabs_a = (C.c_void_p * n)(*d_slabs)
                grads_a = (N.LayerGrads * n)()
                for i_, gp in enumerate(d_grads):
                    for name_, v_ in zip(("g_wq", "g_wk", "g_wv", "g_wo", "g_ln", "g_lnq", "g_lnk", "g_tab_q", "g_tab_k"), gp):
                        setattr(grads_a[i_], name_, v_)
                with self.phase("layer_bwd_reduce"):
                    N.check(lib.satrans_layer_bwd_reduce(n, descs_a, slabs_a, grads_a, C.byref(d_head) if d_head is not None else None,
                                                         self._stream()), "satrans_layer_bwd_reduce")
            if modulated:
                with self.phase("scenario_bwd"):
                    self.scenario_tables_backward(g_tabs)

        # train_step: the next batch's preprocessing forks to the side stream HERE, right behind the last backward kernel and in
        # front of the reduction (the step timeline showed it starting 48 us later when forked behind `finish`, and the next
        # step's first kernel waiting 29 us for it: profiles/r04_step_timeline_*.txt)
        # (ONE fork event for both side streams: every event recorded on the launch stream is a marker its next kernel waits behind)
        fork = None
        if after_layers is not None or (side_tail and d_descs):
            fork = torch.cuda.Event()
            fork.record(torch.cuda.current_stream(self.dev))
        if after_layers is not None:
            after_layers(fork)
        self._last_fork = fork            # (after_step: readers of this step's probabilities start here too)
        # train_step (side_tail): the reduction and the scenario-table backward feed only the flat Adam launch at the very end of
        # the step, while the five touched-row launches that come first need only the last backward kernel's dx.  On a stream of
        # their own (NOT the next-batch stream: queued behind each other the two made the next step wait, which is what the first
        # attempt measured as "45 us slower") they run beside the touched-row chain; the flat Adam waits for `_tail_done`, an
        # event that has long completed by then.
        self._tail_done = None
        if side_tail and d_descs:
            main = torch.cuda.current_stream(self.dev)
            if self._side_tail is None:
                self._side_tail = _shared_stream(self.dev, "tail", lambda: torch.cuda.Stream(self.dev))
            self._side_tail.wait_event(fork)
            with torch.cuda.stream(self._side_tail):
                if clear_late and not precleared:
                    g_clear.zero_()
                finish()
                self._tail_done = torch.cuda.Event()
                self._tail_done.record(self._side_tail)
        else:
            if clear_late and not precleared:
                g_clear.zero_()
            finish()
        self._last_prob = ws["prob"]
        return ws["dact"][cur]

    # ------------------------------------------------------------------------------------------------
    # next-batch preprocessing on a side stream
    # ------------------------------------------------------------------------------------------------
    _PREP_KEYS = ("rows", "sorted_rows", "src", "sid", "order", "seg", "bucket")

    def _prep_key(self, X):
        return (X.data_ptr(), tuple(X.shape), X.dtype, X.stride(0))

    def _can_prepare(self, X, B) -> bool:
        return (self.prefetch and torch.is_tensor(X) and X.is_cuda and X.dim() == 2 and X.is_contiguous()
                and X.dtype in (torch.float32, torch.int32, torch.int64) and not self.multi and self.lazy
                and self._sort_fields is not None and B <= 8192 and self.fuse_gather and not self.force_split)

    def _low_priority_stream(self, name: str = "low_priority"):
        """A HIP stream of the lowest priority the device offers (torch only hands out priorities <= 0, i.e. normal and above),
        created by libsatrans_hip.so - which is linked against the HIP runtime this process already uses - and wrapped for torch:
        when its kernels and the launch stream's become ready together, the launch stream's are dispatched first.  Lives as long
        as the process (kept in the module-wide _STREAMS table, shared by every engine on the device; never destroyed)."""
        def make():
            handle = C.c_void_p()
            with torch.cuda.device(self.dev):
                rc = self.lib.satrans_stream_create_low_priority(C.byref(handle))
            if rc == 0 and handle.value:
                return torch.cuda.ExternalStream(handle.value, device=self.dev)      # (lives as long as the process: _STREAMS)
            return None
        st = _shared_stream(self.dev, name, make)
        if st is not None or name != "low_priority":
            return st
        # no priorities on this device / runtime: an early fork would race the last backward kernel for the CUs - fork behind it
        self.prep_early = False
        return _shared_stream(self.dev, "side", lambda: torch.cuda.Stream(self.dev))

    def _prepare_async(self, X_next, ws_cur_B, fork=None, defer_bucket=False):
        """Everything of a step that depends on nothing but its id matrix - ids -> arena rows, the per-field sort of the rows,
        the scenario bucketing (four launches, ~60 us of mostly idle GPU: one workgroup per field / one workgroup) - for the
        NEXT batch, on a side stream, into the other half of a double buffer.  Called from inside the current step right after
        its last layer kernel has been queued: the side stream starts there, i.e. underneath the current step's small tail
        kernels (reductions, scenario-table backward, touched-row Adam, flat Adam), never underneath the persistent layer
        kernels (they own every CU: a 19-workgroup sort beside them would hold back 19 of their workgroups for its whole
        duration).  Consumed by the next train_step when it is given the SAME tensor, unchanged; anything else discards it."""
        B = X_next.shape[0]
        ws = self.train_workspace(B, 1, False)
        if "prep_alt" not in ws:
            ws["prep_alt"] = {k: torch.empty_like(ws[k]) for k in self._PREP_KEYS}
        alt = ws["prep_alt"]
        main = torch.cuda.current_stream(self.dev)
        if self._side is None:
            self._side = self._low_priority_stream() if self.prep_early else \
                _shared_stream(self.dev, "side", lambda: torch.cuda.Stream(self.dev))
        if fork is None:
            fork = torch.cuda.Event()
            fork.record(main)
        self._side.wait_event(fork)
        with torch.cuda.stream(self._side):
            st = self._stream()
            # ids -> rows and the per-field sort of the rows in ONE launch (the row matrix is written on the way)
            f_, lo_, n_ = self._sort_fields
            N.check(self.lib.satrans_embed_rows_sort_fields(X_next.data_ptr(), N.id_dtype_of(X_next), X_next.stride(0),
                                                            self.cols.data_ptr(), self.row_span.data_ptr(), alt["rows"].data_ptr(),
                                                            B, self.F, f_, lo_, n_, alt["sorted_rows"].data_ptr(),
                                                            alt["src"].data_ptr(), self.status.data_ptr(), st),
                    "satrans_embed_rows_sort_fields(next)")
            # two events: the next step's first launch (the replay) needs the sorted rows only; the bucketing is waited for
            # behind it, in front of the first layer (13 us + a launch gap that the step's start does not wait for)
            sorted_ev = torch.cuda.Event()
            sorted_ev.record(self._side)
            done = None
            if not defer_bucket:      # (owner form: the id exchange goes first, _prepare_owner_async finishes the preparation)
                self._bucket(X_next, alt)
                done = torch.cuda.Event()
                done.record(self._side)
        self._prep = dict(key=self._prep_key(X_next), X=X_next, B=B, done=done, sorted=sorted_ev)

    def _take_prepared(self, X, ws) -> bool:
        """True when the preprocessing of exactly this batch is waiting in the other half of the double buffer: the halves
        swap roles and the launch stream waits for the side stream's event."""
        prep, self._prep = self._prep, None
        if prep is None:
            return False
        if prep["key"] != self._prep_key(X) or prep["B"] != X.shape[0] or "prep_alt" not in ws:
            torch.cuda.current_stream(self.dev).wait_event(prep["done"] or prep["sorted"])      # (the discarded work still owns the alt buffers)
            return False
        alt = ws["prep_alt"]
        for k in self._PREP_KEYS:
            ws[k], alt[k] = alt[k], ws[k]
        # (one event for both, waited for here: 1.141 against 1.130 ms/step, four A/B rounds on one box)
        torch.cuda.current_stream(self.dev).wait_event(prep["sorted"])
        self._prep_rest = prep["done"] or prep["sorted"]      # (waited for by _join_prepared, in front of the first reader of the bucketing)
        return True

    def _join_prepared(self) -> None:
        ev, self._prep_rest = getattr(self, "_prep_rest", None), None
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)

    def train_step(self, X: torch.Tensor, y: torch.Tensor, next_X: Optional[torch.Tensor] = None):
        """One optimizer step (reference models/meta_basemodel.py:310-328 + torch.optim.Adam.step, main.py:343).

        `next_X` (optional): the id matrix the NEXT call will be given - `fit` knows it; its preprocessing then runs on a side
        stream underneath this step's tail (_prepare_async).  A hint only: parameters and optimizer state are the same bits
        with or without it.  One visible difference: a step called WITH `next_X` is pipelined into the next one - its dense
        gradient buffer (the `param.grad` views) is cleared for the next step behind its own flat Adam launch, so `.grad` is not
        meaningful after such a step; a step called without `next_X` leaves its gradients in place until the next step
        (the reference keeps them until the next zero_grad()).

        Three forms of the Adam step, one file each, picked here:
          one rank                         engine_local.py        _train_step_local
          several ranks, row ownership     engine_owner.py        _train_step_owner       (default with torch.distributed)
          several ranks, replicated        engine_replicated.py   _train_step_replicated  (SATRANS_DP_MODE=replicated)
        and the dense SGD / Adagrad / RMSprop step for API parity (_train_step_dense)."""
        from . import parallel
        X = self._prepare_input(X)
        y = y.reshape(-1).to(torch.float32).contiguous()
        B = X.shape[0]
        self._ensure_train_state()
        cfg = self.m._refresh_adam_cfg()              # lr schedulers / edited param_groups take effect at this step
        if cfg.get("kind", "adam") != "adam":
            return self._train_step_dense(X, y, cfg)
        world = parallel.world_size()
        exch = parallel.exchange_enabled()            # several ranks (or one rank made to run its collectives: tests)
        if exch and self.dp_mode == "owner" and self.lazy and self.F_small < self.F:
            return self._train_step_owner(X, y, B, world, self.train_workspace(B, 1, False), next_X)
        if self._owner_world:                          # (a run that switches forms mid-way: bring the replicas together first)
            self.flush_lazy()
            self._owner_world = 0
        if exch:
            return self._train_step_replicated(X, y, B, world, cfg)
        return self._train_step_local(X, y, B, cfg, next_X)

    # ---- pieces the step forms share ------------------------------------------------------------------------------------
    def _sort_rows(self, ws, B, rows, n, out_rows, out_src, touched, per_field=False):
        """Stable sort of arena rows with their positions.  per_field: `rows` is THIS batch's [B, F] row matrix (column f inside
        field f's table) - the only input the one-workgroup-per-field kernel may see; the rank-major concatenation of an
        exchange is not one."""
        lib = self.lib
        if per_field and self._sort_fields is not None and touched is None and n == B * self.F and B <= 8192:
            f_, lo_, n_ = self._sort_fields       # one workgroup per field, one launch
            N.check(lib.satrans_embed_sort_fields(rows.data_ptr(), B, self.F, f_, lo_, n_, out_rows.data_ptr(),
                                                  out_src.data_ptr(), self._stream()), "satrans_embed_sort_fields")
            return
        N.check(lib.satrans_embed_sort(rows.data_ptr(), n, self.total_rows, out_rows.data_ptr(), out_src.data_ptr(),
                                       touched, ws["sort_ws"].data_ptr(), ws["sort_ws"].numel(),
                                       ws["iota"].data_ptr(), self._stream()),
                "satrans_embed_sort")

    def _replay_rows(self, sorted_rows, n, reg, reg_sum):
        """Lazy form: the postponed steps of exactly these rows, up to step adam_t - 1."""
        m, lib = self.m, self.lib
        table = self._table(self.adam_t)
        h = self._hparams(m.l2_reg_embedding)
        N.check(lib.satrans_embed_lazy_replay(m.embedding_arena.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
                                              self.last_step.data_ptr(), self.D, sorted_rows.data_ptr(), n,
                                              self.adam_t - 1, table.data_ptr(), C.byref(h), reg.data_ptr(),
                                              self._stream()), "satrans_embed_lazy_replay")
        if reg_sum is not None:      # (the main-stream replay's partials are summed with the others at the end of the step)
            N.check(lib.satrans_sum_f64(reg.data_ptr(), reg.numel(), reg_sum.data_ptr(), 1, self._stream()),
                    "satrans_sum_f64")

    def _rows_sorted(self, X, ws, B, st):
        """This batch's arena rows (nothing is moved yet) and their per-field sort, on the launch stream."""
        lib, m = self.lib, self.m
        if self._sort_fields is not None and B <= 8192:
            f_, lo_, n_ = self._sort_fields           # ids -> rows + per-field sort, one launch
            with self.phase("embed_sort"):
                N.check(lib.satrans_embed_rows_sort_fields(X.data_ptr(), N.id_dtype_of(X), X.stride(0), self.cols.data_ptr(),
                                                           self.row_span.data_ptr(), ws["rows"].data_ptr(), B, self.F, f_, lo_, n_,
                                                           ws["sorted_rows"].data_ptr(), ws["src"].data_ptr(),
                                                           self.status.data_ptr(), st), "satrans_embed_rows_sort_fields")
        else:
            N.check(lib.satrans_gather_fwd(m.embedding_arena.data_ptr(), self.row_span.data_ptr(), self.cols.data_ptr(), X.data_ptr(),
                                           N.id_dtype_of(X), X.stride(0), B, self.F, self.D, None, ws["rows"].data_ptr(),
                                           self.status.data_ptr(), st), "satrans_gather_fwd(rows)")
            with self.phase("embed_sort"):
                self._sort_rows(ws, B, ws["rows"], B * self.F, ws["sorted_rows"], ws["src"], None, per_field=True)

    def _small_tables_step(self, ws, small_rows, h_emb, st):
        """Dense step over every row of the small tables (their gradient is the dense buffer g_small)."""
        m, lib = self.m, self.lib
        with self.phase("adam_small"):
            N.check(lib.satrans_embed_adam_rows(m.embedding_arena.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
                                                self.last_step.data_ptr(), 0, small_rows, self.D,
                                                self.g_small.data_ptr(), C.byref(h_emb), self.adam_t,
                                                ws["reg_rows"].data_ptr(), st), "satrans_embed_adam_rows")
            N.check(lib.satrans_sum_f64(ws["reg_rows"].data_ptr(), ws["reg_rows"].numel(), self.reg_sum.data_ptr(), 1,
                                        st), "satrans_sum_f64")

    def _train_step_dense(self, X, y, cfg):
        """One step of SGD / Adagrad / RMSprop with the reference's dense semantics (models/meta_basemodel.py:612-640): the
        gradient of EVERY table row (gathered rows + 2 l2 p) is materialised once and one elementwise kernel steps the tables,
        another the remaining parameters.  One sweep over the tables per step - these optimizers are accepted for API parity,
        the Adam path (reference main.py:343) is the tuned one."""
        from . import parallel
        B = X.shape[0]
        world, exch = parallel.world_size(), parallel.exchange_enabled()
        if self._owner_world:                          # (Adam steps in the owner form came before: replicas together first)
            self.flush_lazy()
            self._owner_world = 0
        ws = self.train_workspace(B, 1, False)
        lib, m, D, st = self.lib, self.m, self.D, self._stream()
        l2 = float(m.l2_reg_embedding)
        if l2 > 0:                                    # the step's regulariser term of the logged loss (pre-update weights)
            self.reg_sum += l2 * torch.sum(torch.square(m.embedding_arena.double()))
        gemb = self.backward(X, y, ws)
        n_rows = B * self.F
        rows_t, grads_t = ws["rows"], gemb
        if exch:
            # several ranks (reference semantics as for Adam: per-GPU batches, summed loss): the dense parameters' gradient is
            # all-reduced, the (row, gradient row) lists of all ranks are concatenated rank-major; the dense table gradient is
            # then built from the merged list exactly as on one rank, identically on every rank (the regulariser term once)
            parallel.all_reduce_flat(self.flat_g)
            rows_t = parallel.gather_rows(ws["rows"])
            grads_t, pending = parallel.gather_grad_rows_async(gemb)
            if pending is not None:
                pending.wait()
            n_rows *= world
            if ws.get("_dense_n", 0) < n_rows:
                i32 = dict(dtype=torch.int32, device=self.dev)
                ws["dn_sorted"], ws["dn_src"] = torch.empty(n_rows, **i32), torch.empty(n_rows, **i32)
                ws["dn_iota"] = torch.arange(n_rows, **i32)
                ws["dn_sort_ws"] = torch.empty(int(lib.satrans_embed_sort_workspace_bytes(n_rows, self.total_rows)),
                                               dtype=torch.uint8, device=self.dev)
                ws["_dense_n"] = n_rows
        srt, src = (ws["dn_sorted"], ws["dn_src"]) if exch else (ws["sorted_rows"], ws["src"])
        sort_ws, iota = (ws["dn_sort_ws"], ws["dn_iota"]) if exch else (ws["sort_ws"], ws["iota"])
        N.check(lib.satrans_embed_sort(rows_t.data_ptr(), n_rows, self.total_rows, srt.data_ptr(), src.data_ptr(), None,
                                       sort_ws.data_ptr(), sort_ws.numel(), iota.data_ptr(), st), "satrans_embed_sort")
        self._g_arena.zero_()
        N.check(lib.satrans_embed_grad_dense(m.embedding_arena.data_ptr(), srt.data_ptr(), src.data_ptr(),
                                             n_rows, grads_t.data_ptr(), self.total_rows, D, l2, self._g_arena.data_ptr(), st),
                "satrans_embed_grad_dense")
        kind = {"sgd": 1, "adagrad": 2, "rmsprop": 3}[cfg["kind"]]
        lr, alpha, eps = float(cfg["lr"]), float(cfg.get("alpha", 0.0)), float(cfg.get("eps", 0.0))
        N.check(lib.satrans_optim_flat(kind, m.embedding_arena.data_ptr(), self._g_arena.data_ptr(),
                                       self._opt_state_arena.data_ptr(), m.embedding_arena.numel(), lr, alpha, eps, st),
                "satrans_optim_flat")
        N.check(lib.satrans_optim_flat(kind, m.flat_params.data_ptr(), self.flat_g.data_ptr(), self._opt_state_flat.data_ptr(),
                                       m.flat_params.numel(), lr, alpha, eps, st), "satrans_optim_flat")
        self.adam_t += 1

    def _flat_step(self, h_flat, ws, st):
        lib, m = self.lib, self.m
        N.check(lib.satrans_adam_flat_sum(m.flat_params.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                          self.flat_v.data_ptr(), m.flat_params.numel(), C.byref(h_flat),
                                          ws["reg_partials"].data_ptr(), ws["reg_partials"].numel(),
                                          self.reg_sum.data_ptr(), st), "satrans_adam_flat_sum")

    def _note_lr(self, lr: float) -> None:
        """Called by every Adam step right after it has advanced `adam_t`: step adam_t (and, until further notice, every later
        one) is taken with learning rate `lr`.  When the rate differs from the one the per-step table holds for this step (an LR
        scheduler, an edited param_group) the rows from this step on are rewritten - the rows of earlier, possibly still
        postponed, steps keep the rate they were taken with, so a rate change needs no flush."""
        hist = getattr(self, "_lr_hist", None)
        if hist is None:
            hist = self._lr_hist = {}                   # step -> rate, only where it changed: {first step of a segment: rate}
        prev = getattr(self, "_table_lr", None)
        if prev is None or lr != prev:
            hist[self.adam_t] = lr
            starts = getattr(self, "_lr_starts", None)
            if starts is not None and (not starts or self.adam_t > starts[-1]):
                starts.append(self.adam_t)          # (steps only grow: the sorted key list stays sorted)
            else:
                self._lr_starts = None
            self._table_lr = lr
            if self._hp_table is not None:
                d = getattr(self, "_table_dirty", None)
                self._table_dirty = self.adam_t if d is None else min(d, self.adam_t)

    def _prune_lr_history(self) -> None:
        """After a flush every row is current as of step adam_t: no step before it is ever replayed again, so the rate changes
        before the one in force at adam_t can go - a scheduler that changes the rate every step leaves at most
        `flush_every` + 1 entries instead of one per step of the run."""
        hist = getattr(self, "_lr_hist", None)
        if not hist or len(hist) < 2:
            return
        past = [s_ for s_ in hist if s_ <= self.adam_t]
        if len(past) > 1:
            keep = max(past)
            self._lr_hist = {s_: r for s_, r in hist.items() if s_ >= keep}
            self._lr_starts = None

    def _rates(self, lo: int, hi: int):
        """Learning rate of the steps [lo, hi) as an fp64 array: the rate of the last change at or before each step (steps before
        the first recorded change - a resumed run - take the first recorded rate; they are never replayed).  The history is
        kept in step order (every `_note_lr` appends), so the segments that matter are found by bisection: a scheduler that
        changes the rate every step costs O(changes inside [lo, hi)) per call, not O(all changes so far)."""
        import bisect
        import numpy as np
        hist = getattr(self, "_lr_hist", None) or {0: self.m._adam_cfg["lr"]}
        starts = getattr(self, "_lr_starts", None)
        if starts is None or len(starts) != len(hist):
            starts = self._lr_starts = sorted(hist)
        out = np.empty(max(0, hi - lo), dtype=np.float64)
        i0 = max(0, bisect.bisect_right(starts, lo) - 1)
        for i in range(i0, len(starts)):
            st = starts[i]
            if st >= hi:
                break
            a_, b_ = max(lo, st if i else 0), (starts[i + 1] if i + 1 < len(starts) else hi)
            if b_ > a_:
                out[a_ - lo:min(b_, hi) - lo] = hist[st]
        return out

    def _table(self, upto: int) -> torch.Tensor:
        """Per-step Adam constants for the replay kernels: row s = (fp32(lr_s / (1 - beta1^s)), 1 / fp32(sqrt(1 - beta2^s))),
        exactly what `_hparams` hands the step kernels at step s (same Python arithmetic for the powers; the division and the
        fp32 rounding are IEEE operations in numpy as in Python), lr_s = the rate step s was / will be taken with (`_note_lr`)."""
        import numpy as np
        betas = tuple(self.m._adam_cfg["betas"])
        dirty = getattr(self, "_table_dirty", None)
        if self._hp_table is None or self._hp_cfg != betas or self._hp_table.shape[0] <= upto:
            cap = max(4096, 2 * (upto + 1))
            b1, b2 = betas
            f32 = lambda x: float(np.float32(x))          # the value the fp32 kernels (and torch's fp32 step) see
            self._hp_bc1 = np.array([1.0] + [1.0 - b1 ** s_ for s_ in range(1, cap)], dtype=np.float64)
            col1 = np.array([1.0] + [1.0 / f32(math.sqrt(1.0 - b2 ** s_)) for s_ in range(1, cap)], dtype=np.float64)
            col0 = np.float32(self._rates(0, cap) / self._hp_bc1).astype(np.float64)
            col0[0] = 0.0
            self._hp_table = torch.from_numpy(np.stack([col0, col1], axis=1)).to(self.dev).contiguous()
            self._hp_cfg = betas
            self._table_dirty = None
        elif dirty is not None:
            cap = self._hp_table.shape[0]
            col0 = np.float32(self._rates(dirty, cap) / self._hp_bc1[dirty:]).astype(np.float64)
            if dirty == 0:
                col0[0] = 0.0
            self._hp_table[dirty:, 0].copy_(torch.from_numpy(col0))
            self._table_dirty = None
        return self._hp_table

    def flush_lazy(self, sync: bool = True):
        """Bring every table row (owner form: every row this rank owns) to the current step - a no-op when nothing is pending -
        and, with `sync`, make the replicas of all ranks identical again (owner form: COLLECTIVE - slice broadcasts; every rank
        reaches the flush points that `fit` / `predict` / `evaluate` contain together, as it reaches the steps together).
        `sync="local"` is for entry points a single rank may call on its own (state_dict, optimizer_state_dict, `.to()`): with
        stale replicas it raises instead of starting a collective the other ranks never join (a rank-0-only
        checkpoint in the middle of an epoch would hang); `model.synchronize()` on every rank first makes it legal."""
        self._join_flat()
        self._join_roll()
        if self.lazy and self._lazy_pending:
            m = self.m
            h = self._hparams(m.l2_reg_embedding) if self.adam_t > 0 else None
            st = self._stream()
            table = self._table(self.adam_t)
            with self.phase("lazy_flush"):
                self._flush_launches(m, h, st, table)
            self._lazy_pending = False
            self._since_flush = 0
            self.flush_count = getattr(self, "flush_count", 0) + 1
            self._prune_lr_history()
        if sync and self._replicas_stale:
            from . import parallel
            if sync == "local" and parallel.world_size() > 1:
                raise RuntimeError("the embedding tables of this rank are stale outside the slice it owns (data-parallel owner "
                                   "form, mid-epoch): call model.synchronize() on EVERY rank first - it is a collective - or read "
                                   "the model at an epoch boundary (fit() leaves the replicas identical)")
            self._sync_replicas()

    def _flush_launches(self, m, h, st, table):
        lo, hi = 0, self.total_rows
        if self._owner_world:          # only the rows this rank steps: its slice of the large tables (small tables are never lazy)
            from . import parallel
            b = self._owner_ranges(self._owner_world)
            lo, hi = b[parallel.rank()], b[parallel.rank() + 1]
            if hi <= lo:
                return
        N.check(self.lib.satrans_embed_lazy_flush(m.embedding_arena[lo:].data_ptr(), self.adam_m[lo:].data_ptr(),
                                                  self.adam_v[lo:].data_ptr(), self.last_step[lo:].data_ptr(), hi - lo,
                                                  self.D, self.adam_t, table.data_ptr(), C.byref(h), 0,
                                                  self._flush_reg.data_ptr(), st), "satrans_embed_lazy_flush")
        N.check(self.lib.satrans_sum_f64(self._flush_reg.data_ptr(), self._flush_reg.numel(), self.reg_sum.data_ptr(),
                                         1, st), "satrans_sum_f64")

    def _join_roll(self):
        """The launch stream waits for the rolling flush of the previous step (it writes table rows and their `last_step`)."""
        if self._roll_done is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._roll_done)
            self._roll_done = None

    def _roll_flush(self, fork) -> bool:
        """Rolling form of the periodic flush (one rank, lazy): slice `adam_t mod flush_every` of the rows is brought up to step
        adam_t - 1 on a lowest-priority stream that starts at `fork` (behind the last backward kernel of step adam_t).  The target
        is the step BEFORE the one whose touched-row launch runs beside it: the rows of that launch were replayed to adam_t - 1 at
        the head of the step, so the flush kernel skips them (`last_step >= target`: neither read nor written) and the two never
        meet on a row.  The next reader or writer of any row - the next step's replay, flush_lazy() - waits for `_roll_done`.
        Same row-steps as the one-launch flush, each executed once, so the tables stay the same bits."""
        target = self.adam_t - 1
        if target < 1 or fork is None:
            return False
        if self._roll is None:
            self._roll = self._low_priority_stream("roll")
            if self._roll is None:
                self.rolling_flush = False
                return False
            self._roll_reg = torch.zeros(self._flush_reg.numel(), dtype=torch.float64, device=self.dev)
        m, n_sl = self.m, self.flush_every
        s = self.adam_t % n_sl
        lo, hi = self.total_rows * s // n_sl, self.total_rows * (s + 1) // n_sl
        if hi <= lo:
            return True
        h = self._hparams(m.l2_reg_embedding)
        table = self._table(self.adam_t)               # (built / patched on the launch stream, in front of `fork`)
        self._roll.wait_event(fork)
        with torch.cuda.stream(self._roll):
            st = self._stream()
            with self.phase("lazy_flush_roll"):
                N.check(self.lib.satrans_embed_lazy_flush(m.embedding_arena[lo:].data_ptr(), self.adam_m[lo:].data_ptr(),
                                                          self.adam_v[lo:].data_ptr(), self.last_step[lo:].data_ptr(), hi - lo,
                                                          self.D, target, table.data_ptr(), C.byref(h), 0,
                                                          self._roll_reg.data_ptr(), st), "satrans_embed_lazy_flush")
            N.check(self.lib.satrans_sum_f64(self._roll_reg.data_ptr(), self._roll_reg.numel(), self.reg_roll.data_ptr(), 1, st),
                    "satrans_sum_f64")
            self._roll_done = torch.cuda.Event()
            self._roll_done.record(self._roll)
        return True

    # ------------------------------------------------------------------------------------------------
    # inspection for the parity tests: one forward+backward, gradients by state_dict key (dense tables)
    # ------------------------------------------------------------------------------------------------
    def loss_and_grads(self, X: torch.Tensor, y: torch.Tensor):
        self.flush_lazy()
        X = self._prepare_input(X)
        y = y.reshape(-1).to(torch.float32).contiguous()
        B = X.shape[0]
        self._ensure_train_state()
        n_rows = B * self.F
        ws = self.train_workspace(B, 1)
        lib, st, m, D = self.lib, self._stream(), self.m, self.D
        self.loss_sum.zero_()
        gemb = self.backward(X, y, ws)
        N.check(lib.satrans_embed_sort(ws["rows"].data_ptr(), n_rows, self.total_rows, ws["sorted_rows"].data_ptr(),
                                       ws["src"].data_ptr(), ws["touched"].data_ptr(), ws["sort_ws"].data_ptr(),
                                       ws["sort_ws"].numel(), None, st), "satrans_embed_sort")
        g_arena = torch.zeros_like(m.embedding_arena)
        N.check(lib.satrans_embed_grad_dense(m.embedding_arena.data_ptr(), ws["sorted_rows"].data_ptr(),
                                             ws["src"].data_ptr(), n_rows, gemb.data_ptr(), self.total_rows, D,
                                             float(m.l2_reg_embedding), g_arena.data_ptr(), st),
                "satrans_embed_grad_dense")
        grads = {}
        for name, (off, rows) in m._table_rows.items():
            grads[f"embedding_dict.{name}.weight"] = g_arena[off:off + rows].clone()
        for name, p in m._trainable_flat().items():
            off, cnt = m._flat_slices[name]
            grads[name] = self.flat_g[off:off + cnt].view(p.shape).clone()
        bce = float(self.loss_sum.item())
        reg = float(m.get_regularization_loss().item())
        self.raise_if_bad_ids()
        return bce, reg, grads
