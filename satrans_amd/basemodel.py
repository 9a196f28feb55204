"""Training / evaluation harness with the reference's `BaseModel` surface.

Mirrors the public behaviour of reference models/meta_basemodel.py (itself a fork of deepctr-torch's
BaseModel): constructor arguments and parameter creation order (:125-188), `compile` (:598-610),
`fit` (:200-385), `evaluate` (:387-399), `predict` (:401-517), metric names (:655-671) and the
state_dict key names.  What differs is underneath:

  * all embedding tables live back to back in ONE fp32 arena in HBM (`self.embedding_arena`), the
    `embedding_dict.<name>.weight` parameters are views into it, so the fused gather reads one
    address space and the optimizer streams one buffer;
  * every other trainable tensor is a view into one flat buffer (`self.flat_params`) stepped by one
    fused Adam launch;
  * `fit` keeps the packed dataset resident in HBM, never calls autograd for the hot path, and only
    reads losses back once per epoch (or per step when verbose metrics are requested, as the
    reference does);
  * forward / backward / optimizer are HIP kernels reached through satrans_amd.native.
"""
from __future__ import annotations

import time
import warnings
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .callbacks import CallbackList, History
from .inputs import DenseFeat, SparseFeat, VarLenSparseFeat, build_input_features, split_columns

# _READ_ONLY_FILTER: torch.from_numpy on a read-only memory map of an HDF5 column (h5lite) warns that the array "is not
# writable"; the upload only reads it.  Installed ONCE, here, for that one message: warnings.catch_warnings() swaps the
# process-global filter list and is not thread-safe, and the upload runs in a worker thread beside user callbacks.
warnings.filterwarnings("ignore", message="The given NumPy array is not writable", category=UserWarning)


# tables of at most this many rows (and this many in total) form the "small" class of the optimizer
SMALL_TABLE_ROWS = 16384
SMALL_TABLE_TOTAL_ROWS = 65536


def make_embedding_tables(feature_columns, init_std=0.0001, linear=False, skip_init=False) -> nn.ModuleDict:
    """One table per SparseFeat, keyed by `embedding_name`; N(0, init_std) after torch's default N(0,1)
    draw, exactly the sequence of generator draws of reference models/meta_basemodel.py:95-121."""
    sparse, _, varlen = split_columns(feature_columns)
    tables = nn.ModuleDict()
    for col in sparse + varlen:
        tables[col.embedding_name] = nn.Embedding(col.vocabulary_size, 1 if linear else col.embedding_dim)
    if not skip_init:
        for table in tables.values():
            nn.init.normal_(table.weight, mean=0, std=init_std)
    return tables


class _LinearTables(nn.Module):
    """The order-1 ("linear") part of deepctr models.  SATrans never evaluates it
    (reference models/satrans.py:197-256) but it owns state_dict keys `linear_model.*` and consumes
    generator draws during construction (models/meta_basemodel.py:34-61), so it is kept as storage."""

    def __init__(self, feature_columns, init_std=0.0001):
        super().__init__()
        _, dense, _ = split_columns(feature_columns)
        self.embedding_dict = make_embedding_tables(feature_columns, init_std, linear=True)
        for table in self.embedding_dict.values():          # the reference initialises these a second time (:55-56)
            nn.init.normal_(table.weight, mean=0, std=init_std)
        if dense:
            self.weight = nn.Parameter(torch.empty(sum(c.dimension for c in dense), 1))
            nn.init.normal_(self.weight, mean=0, std=init_std)


class _Prediction(nn.Module):
    """Holds the `out.bias` parameter of deepctr's PredictionLayer (unused by SATrans.forward, which applies
    torch.sigmoid directly at models/satrans.py:255)."""

    def __init__(self, task="binary"):
        super().__init__()
        if task not in ("binary", "multiclass", "regression"):
            raise ValueError("task must be binary,multiclass or regression")
        self.task = task
        self.bias = nn.Parameter(torch.zeros((1,)))


class BaseModel(nn.Module):
    def __init__(self, linear_feature_columns, dnn_feature_columns, l2_reg_linear=1e-5, l2_reg_embedding=1e-5,
                 init_std=0.0001, seed=1024, task='binary', device='cpu', gpus=None, flag=None):
        super().__init__()
        self.flag = flag
        self.embedding_dim = dnn_feature_columns[0].embedding_dim
        torch.manual_seed(int(seed))                        # main.py:43 passes the seed as a string
        self.dnn_feature_columns = dnn_feature_columns
        self.linear_feature_columns = linear_feature_columns
        self.device = device
        self.gpus = gpus
        if gpus and str(self.gpus[0]) not in self.device:
            raise ValueError("`gpus[0]` should be the same gpu with `device`")
        if gpus and len(gpus) > 1:
            # reference: single-process torch.nn.DataParallel over `gpus` (models/meta_basemodel.py:272-275).  Here data
            # parallelism is one PROCESS per GPU (torchrun + RCCL, satrans_amd/parallel.py) with the same semantics
            # (batch_size per GPU, summed gradients); a model constructed in one process drives one GPU.
            import warnings
            warnings.warn("gpus=%s: this build is data-parallel with one process per GPU (launch with torchrun; fit() "
                          "then shards by rank). Inside one process only `device` is used." % (list(gpus),))
        self.l2_reg_embedding = float(l2_reg_embedding)
        self.l2_reg_linear = float(l2_reg_linear)

        self.feature_index = build_input_features(list(linear_feature_columns) + list(dnn_feature_columns))
        self.embedding_dict = make_embedding_tables(dnn_feature_columns, init_std,
                                                    skip_init=bool(flag) and 'noembinit' in flag)
        self.linear_model = _LinearTables(linear_feature_columns, init_std)
        self.out = _Prediction(task)

        self.reg_loss = torch.zeros((1,))
        self.aux_loss = torch.zeros((1,))
        self.history = History()
        self.stop_training = False
        self._engine = None
        self.embedding_arena: Optional[torch.Tensor] = None
        self.flat_params: Optional[torch.Tensor] = None
        # parameters for callbacks (same attribute names as the reference, used by keras-style callbacks)
        self._is_graph_network = True
        self._ckpt_saved_epoch = False

    # ------------------------------------------------------------------------------------------
    # storage: one arena for the tables, one flat buffer for everything else that trains
    # ------------------------------------------------------------------------------------------
    @property
    def embedding_size(self):
        sparse, _, varlen = split_columns(self.dnn_feature_columns)
        dims = {c.embedding_dim for c in sparse + varlen}
        if len(dims) > 1:
            raise ValueError("embedding_dim of SparseFeat and VarlenSparseFeat must be same in this model!")
        return list(dims)[0]

    def _trainable_flat(self) -> "OrderedDict[str, nn.Parameter]":
        """Parameters stepped by the fused flat Adam (subclasses define which receive gradients)."""
        return OrderedDict()

    def _table_order(self) -> List[str]:
        sparse, _, _ = split_columns(self.dnn_feature_columns)
        names, seen = [], set()
        for col in sparse:
            if col.embedding_name not in seen:
                seen.add(col.embedding_name)
                names.append(col.embedding_name)
        return names

    def _rebind_storage(self):
        """(Re)build the arena and the flat buffer on the parameters' current device and turn the
        parameters into views of them.  Called after construction and after every `.to()`."""
        names = self._table_order()
        # Arena order: small tables first (stable), then the large ones.  The optimizer treats the two classes
        # differently (satrans_amd/engine.py: dense all-reduced gradient + dense step for the small tables, sorted
        # (row, gradient) lists for the large ones) and relies on "arena row < _arena_small_rows <=> small table".
        # state_dict keys and values do not depend on the order: the parameters are views at their offsets.
        import os
        limit = int(os.environ.get("SATRANS_SMALL_TABLE_ROWS", SMALL_TABLE_ROWS))    # (tests move the class boundary)
        small, budget = [], SMALL_TABLE_TOTAL_ROWS
        for n in names:
            rows = self.embedding_dict[n].weight.shape[0]
            if rows <= limit and rows <= budget:
                small.append(n)
                budget -= rows
        names = small + [n for n in names if n not in small]
        tables = [self.embedding_dict[n].weight for n in names]
        dev = tables[0].device
        arena = torch.cat([t.data.reshape(-1, t.shape[1]) for t in tables], dim=0).contiguous()
        off = 0
        self._table_rows = OrderedDict()
        for n, t in zip(names, tables):
            rows = t.shape[0]
            t.data = arena[off:off + rows]
            self._table_rows[n] = (off, rows)
            off += rows
        self._arena_small_rows = sum(self.embedding_dict[n].weight.shape[0] for n in small)
        self.embedding_arena = arena
        flat = self._trainable_flat()
        if flat:
            buf = torch.cat([p.data.reshape(-1) for p in flat.values()]).contiguous().to(dev)
            off = 0
            self._flat_slices = OrderedDict()
            for n, p in flat.items():
                cnt = p.numel()
                p.data = buf[off:off + cnt].view(p.shape)
                self._flat_slices[n] = (off, cnt)
                off += cnt
            self.flat_params = buf
        self._engine = None

    def _apply(self, fn, *args, **kwargs):
        # A device / dtype move rebuilds the arena and the engine that drives it.  The engine is the only holder of the
        # optimizer state (Adam moments, step count, dropout step): it is taken out first and handed to the next engine,
        # so `.to()` in the middle of training continues the run as torch.optim.Adam would (its state follows the
        # parameters), instead of silently restarting Adam from step 0.
        eng = getattr(self, "_engine", None)
        carried = getattr(self, "_pending_opt_state", None)
        if eng is not None:
            eng.flush_lazy(sync="local")          # postponed optimizer steps must land before the storage is rebuilt
            if eng.adam_t > 0:
                carried = eng.optimizer_state()
        out = super()._apply(fn, *args, **kwargs)
        if self.embedding_arena is not None:
            self._rebind_storage()
            self._pending_opt_state = carried
        return out

    # ------------------------------------------------------------------------------------------
    # optimizer state (resume): the torch optimizer handed to compile() is only read for its hyper-parameters, the
    # moments live in the engine.  These two calls are the counterpart of optimizer.state_dict()/load_state_dict().
    # ------------------------------------------------------------------------------------------
    def optimizer_state_dict(self) -> dict:
        """Adam state of every trained tensor by state_dict key (`exp_avg`, `exp_avg_sq`: CPU tensors) plus the step
        count and the dropout step; {} before the first step."""
        eng = getattr(self, "_engine", None)
        if eng is None or eng.adam_t == 0:
            pend = getattr(self, "_pending_opt_state", None)
            if pend is None:
                return {}
            raw = pend
        else:
            eng.flush_lazy(sync="local")
            raw = eng.optimizer_state()
        kind = raw.get("kind", "adam")
        out = {"kind": kind, "step": int(raw["adam_t"]), "drop_step": int(raw["drop_step"]), "state": {}}
        # torch's own state keys: Adam `exp_avg` / `exp_avg_sq`, Adagrad `sum`, RMSprop `square_avg`, plain SGD none
        fields = {"adam": (("exp_avg", "adam_m", "flat_m"), ("exp_avg_sq", "adam_v", "flat_v")),
                  "adagrad": (("sum", "acc_arena", "acc_flat"),), "rmsprop": (("square_avg", "acc_arena", "acc_flat"),),
                  "sgd": ()}[kind]
        for name, (off, rows) in self._table_rows.items():
            out["state"][f"embedding_dict.{name}.weight"] = {
                key: raw[arena][off:off + rows].detach().cpu().clone() for key, arena, _ in fields}
        for name, p in self._trainable_flat().items():
            off, cnt = self._flat_slices[name]
            out["state"][name] = {key: raw[flat][off:off + cnt].view(p.shape).detach().cpu().clone()
                                  for key, _, flat in fields}
        return out

    def load_optimizer_state_dict(self, sd: dict) -> None:
        """Inverse of `optimizer_state_dict` (call after `load_state_dict` and `compile`, on the device the run continues on)."""
        if not sd:
            return
        dev = self.embedding_arena.device
        kind = sd.get("kind", "adam")
        fields = {"adam": (("exp_avg", "adam_m", "flat_m"), ("exp_avg_sq", "adam_v", "flat_v")),
                  "adagrad": (("sum", "acc_arena", "acc_flat"),), "rmsprop": (("square_avg", "acc_arena", "acc_flat"),),
                  "sgd": (("", "acc_arena", "acc_flat"),)}[kind]
        raw = {"kind": kind, "adam_t": int(sd["step"]), "drop_step": int(sd.get("drop_step", 0))}
        for _, arena, flat in fields:
            raw[arena] = torch.zeros_like(self.embedding_arena)
            raw[flat] = torch.zeros_like(self.flat_params)
        for name, (off, rows) in self._table_rows.items():
            st = sd["state"][f"embedding_dict.{name}.weight"]
            for key, arena, _ in fields:
                if key:
                    raw[arena][off:off + rows] = st[key].to(dev)
        for name, p in self._trainable_flat().items():
            off, cnt = self._flat_slices[name]
            st = sd["state"][name]
            for key, _, flat in fields:
                if key:
                    raw[flat][off:off + cnt] = st[key].to(dev).reshape(-1)
        eng = getattr(self, "_engine", None)
        if eng is not None:
            eng.flush_lazy()
            eng.load_optimizer_state(raw)
        else:
            self._pending_opt_state = raw

    def _refresh_adam_cfg(self):
        """Re-read lr / betas / eps from the torch optimizer given to compile() (LR schedulers and manual edits of
        `param_groups` take effect at the next step, as they do for torch.optim.Adam).  Steps that are still postponed
        (lazy-exact form) belong to the old values."""
        opt = getattr(self, "optim", None)
        if isinstance(opt, torch.optim.Optimizer):
            new = self._read_optimizer(opt)
            if new != self._adam_cfg:
                old = self._adam_cfg
                # A learning-rate change alone (what an LR scheduler does every step) needs no flush: the replay of postponed
                # steps reads the rate each step was taken with from the engine's per-step table (engine._table).  Anything
                # else (betas, eps, another optimizer) applies the postponed steps under the old values first.
                only_lr = new.get("kind") == "adam" and old.get("kind") == "adam" and \
                    {k: v for k, v in new.items() if k != "lr"} == {k: v for k, v in old.items() if k != "lr"}
                if not only_lr:
                    eng = getattr(self, "_engine", None)
                    if eng is not None:
                        eng.flush_lazy(sync=False)      # (the rows this rank steps; replicas of other owners' rows are not read)
                self._adam_cfg = new
        return self._adam_cfg

    # ------------------------------------------------------------------------------------------
    # compile
    # ------------------------------------------------------------------------------------------
    def compile(self, optimizer, loss=None, metrics=None):
        """Same arguments as the reference (models/meta_basemodel.py:598-610).  The optimizer may be the
        string "adam" or a `torch.optim.Adam` instance (what main.py:343 passes); its hyper-parameters are
        read and the step itself runs as fused HIP kernels.  Other optimizers are not built."""
        self.metrics_names = ["loss"]
        self.optim = optimizer
        self._adam_cfg = self._read_optimizer(optimizer)
        self.loss_func = self._get_loss_func(loss)
        self.metrics = self._get_metrics(metrics)

    @staticmethod
    def _read_optimizer(optimizer):
        """-> dict(kind, lr, ...).  Strings as the reference resolves them (models/meta_basemodel.py:612-640: "sgd" lr 0.01,
        "adam" lr 0.001, "adagrad" lr 0.01, "rmsprop"), or the matching torch.optim instance (its hyper-parameters are read,
        the step itself runs as HIP kernels).  Adam has the lazy-exact table kernels; the others take one dense elementwise
        sweep over the tables per step (reference semantics: dense gradients, every row moves)."""
        if isinstance(optimizer, str):
            table = {"adam": dict(kind="adam", lr=1e-3, betas=(0.9, 0.999), eps=1e-8),
                     "sgd": dict(kind="sgd", lr=0.01),
                     "adagrad": dict(kind="adagrad", lr=0.01, eps=1e-10),
                     "rmsprop": dict(kind="rmsprop", lr=0.01, alpha=0.99, eps=1e-8)}
            if optimizer not in table:
                raise NotImplementedError(optimizer)
            return table[optimizer]
        if len(getattr(optimizer, "param_groups", [0])) != 1:
            raise NotImplementedError("optimizers with several parameter groups")
        g = optimizer.param_groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("maximize", False):
            raise NotImplementedError("weight_decay / maximize")
        if isinstance(optimizer, torch.optim.Adam):
            if g.get("amsgrad", False):
                raise NotImplementedError("Adam with amsgrad")
            return dict(kind="adam", lr=float(g["lr"]), betas=tuple(float(b) for b in g["betas"]), eps=float(g["eps"]))
        if isinstance(optimizer, torch.optim.SGD):
            if g.get("momentum", 0) != 0 or g.get("nesterov", False) or g.get("dampening", 0) != 0:
                raise NotImplementedError("SGD with momentum / nesterov / dampening")
            return dict(kind="sgd", lr=float(g["lr"]))
        if isinstance(optimizer, torch.optim.Adagrad):
            if g.get("lr_decay", 0) != 0 or g.get("initial_accumulator_value", 0) != 0:
                raise NotImplementedError("Adagrad with lr_decay / initial_accumulator_value")
            return dict(kind="adagrad", lr=float(g["lr"]), eps=float(g["eps"]))
        if isinstance(optimizer, torch.optim.RMSprop):
            if g.get("momentum", 0) != 0 or g.get("centered", False):
                raise NotImplementedError("RMSprop with momentum / centered")
            return dict(kind="rmsprop", lr=float(g["lr"]), alpha=float(g["alpha"]), eps=float(g["eps"]))
        raise NotImplementedError(f"optimizer {type(optimizer).__name__}: Adam, SGD, Adagrad and RMSprop have HIP steps")

    @staticmethod
    def _get_loss_func(loss):
        """"binary_crossentropy" / "mse" / "mae" (models/meta_basemodel.py:642-653; summed over the batch in fit)."""
        if loss is None or loss == "binary_crossentropy":
            return "binary_crossentropy"
        if loss in ("mse", "mae"):
            return loss
        raise NotImplementedError(str(loss))

    def _get_metrics(self, metrics, set_eps=False):
        from sklearn.metrics import accuracy_score, log_loss, mean_squared_error, roc_auc_score
        table = OrderedDict()
        for metric in (metrics or []):
            if metric in ("binary_crossentropy", "logloss"):
                table[metric] = log_loss
            if metric == "auc":
                table[metric] = roc_auc_score
            if metric == "mse":
                table[metric] = mean_squared_error
            if metric in ("accuracy", "acc"):
                table[metric] = lambda y_true, y_pred: accuracy_score(y_true, np.where(y_pred > 0.5, 1, 0))
            self.metrics_names.append(metric)
        return table

    # ------------------------------------------------------------------------------------------
    # input packing
    # ------------------------------------------------------------------------------------------
    def _pack(self, x) -> np.ndarray:
        """dict / list of per-feature arrays -> one [N, C] matrix in `feature_index` column order
        (models/meta_basemodel.py:221-222,257-264)."""
        if isinstance(x, dict):
            x = [x[name] for name in self.feature_index]
        cols = [np.asarray(a) for a in x]
        cols = [c.reshape(-1, 1) if c.ndim == 1 else c for c in cols]
        if any(np.issubdtype(c.dtype, np.integer) and c.size and int(c.max()) >= (1 << 24) for c in cols) and \
                any(not np.issubdtype(c.dtype, np.integer) for c in cols):
            cols = [c.astype(np.float64) for c in cols]                  # ids beyond 2**24 next to float columns: exact in fp64
        return np.concatenate(cols, axis=-1)

    def _columns(self, x) -> List[np.ndarray]:
        """dict / list of per-feature arrays -> the [N, w] column blocks in `feature_index` order, nothing concatenated."""
        if isinstance(x, dict):
            x = [x[name] for name in self.feature_index]
        cols = [np.asarray(a) for a in x]
        return [c.reshape(-1, 1) if c.ndim == 1 else c for c in cols]

    def _device_matrix_from_columns(self, cols: List[np.ndarray], lo: int = 0, hi: Optional[int] = None):
        """The resident [N, C] id matrix assembled ON the device: every column block is uploaded as it is (integers stay
        integers on the wire) and cast into its columns of the fp32 matrix there - the values `_pack` + `astype(float32)`
        produce (ids below 2**24 are exact in fp32, float columns are rounded to nearest either way) without the host-side
        concatenate / cast passes over the whole dataset (the reference builds that matrix on the host:
        meta_basemodel.py:257-264).  Vocabularies of 2**24 and above keep the host path (`_host_matrices`: int64 ids + dense
        block)."""
        sparse, _, _ = split_columns(self.dnn_feature_columns)
        if max(c.vocabulary_size for c in sparse) >= (1 << 24):
            return self._to_device_matrix(np.concatenate([c[lo:hi] for c in cols], axis=-1))
        n = (cols[0].shape[0] if hi is None else hi) - lo
        data = torch.empty(n, sum(c.shape[1] for c in cols), dtype=torch.float32, device=self.device)
        j = 0
        for c in cols:
            w = c.shape[1]
            block = np.ascontiguousarray(c[lo:hi])
            if block.dtype == np.float64 or not (np.issubdtype(block.dtype, np.integer) or np.issubdtype(block.dtype, np.floating)):
                block = block.astype(np.float32)                  # (numpy's rounding, as the host path)
            elif block.dtype.kind == "u" and block.dtype.itemsize > 1:
                block = block.astype(np.int64)                    # (torch has no wide unsigned tensors to upload)
            data[:, j:j + w] = torch.from_numpy(block).to(self.device)      # (read-only memory maps: _READ_ONLY_FILTER above)
            j += w
        return data

    def _to_device_matrix(self, packed: np.ndarray) -> torch.Tensor:
        """The reference ships ids as fp32 (exact below 2**24, models/meta_basemodel.py:311).  That layout is kept
        while every vocabulary is below 2**24.  Above it an id would not survive the round trip through fp32 (the
        reference silently gathers the wrong row there): the matrix then travels as int64 and the kernels read the ids
        as integers (`SATRANS_ID_I64`).  Dense features next to such a vocabulary would need a separate float block,
        which no reference dataset calls for."""
        ids, dense_block = self._host_matrices(packed)
        ids_d = torch.from_numpy(ids).to(self.device)
        if dense_block is None:
            return ids_d
        from .inputs import PackedInput
        return PackedInput(ids_d, torch.from_numpy(dense_block).to(self.device))

    def _host_matrices(self, packed: np.ndarray):
        """-> (ids matrix, dense block or None) on the host: fp32 in the reference's layout while every vocabulary is below
        2**24, else int64 ids plus - when the model has DenseFeat columns - their float block (inputs.PackedInput)."""
        sparse, dense, _ = split_columns(self.dnn_feature_columns)
        if max(c.vocabulary_size for c in sparse) >= (1 << 24):
            if not np.issubdtype(packed.dtype, np.integer) and packed.dtype == np.float32 and np.abs(packed).max() >= (1 << 24):
                raise ValueError("ids of 2**24 and above arrived as float32: they are already rounded; pass integers")
            block = None
            if dense:
                cols = []
                for c in dense:
                    lo, hi = self.feature_index[c.name]
                    cols += list(range(lo, hi))
                block = np.ascontiguousarray(packed[:, cols], dtype=np.float32)
                packed = packed.copy()
                packed[:, cols] = 0                                      # (unused as ids; keeps the cast below finite)
            return np.ascontiguousarray(packed.astype(np.int64)), block
        return np.ascontiguousarray(packed, dtype=np.float32), None

    # ------------------------------------------------------------------------------------------
    # fit / evaluate / predict
    # ------------------------------------------------------------------------------------------
    def fit(self, x=None, y=None, batch_size=None, epochs=1, valid_cnt_per_epoch=1, verbose=1, initial_epoch=0,
            validation_split=0., validation_data=None, shuffle=True, callbacks=None):
        """Train.  Arguments and return value as reference models/meta_basemodel.py:200-385."""
        engine = self._require_engine()
        if isinstance(x, dict) and getattr(self, "domain_column_list", None):
            self.domain_id_offset = np.asarray(x[self.domain_column_list[0]]).min()
        cols = self._columns(x)                               # [N, w] blocks; concatenated on the host only when streamed
        n_cols = sum(c.shape[1] for c in cols)
        y = np.asarray(y, dtype=np.float32).reshape(-1)

        do_validation, val_x, val_y = False, None, []
        if validation_data:
            if len(validation_data) not in (2, 3):
                raise ValueError("`validation_data` must be (x_val, y_val) or (x_val, y_val, val_sample_weights); "
                                 "received %s" % (validation_data,))
            do_validation = True
            val_x, val_y = self._pack(validation_data[0]), np.asarray(validation_data[1])
        elif validation_split and 0. < validation_split < 1.:
            do_validation = True
            split_at = int(cols[0].shape[0] * (1. - validation_split))
            val_x = np.concatenate([c[split_at:] for c in cols], axis=-1)
            cols = [c[:split_at] for c in cols]
            y, val_y = y[:split_at], y[split_at:]

        if batch_size is None:
            batch_size = 256
        sample_num = cols[0].shape[0]
        self._check_ranks_agree(sample_num, batch_size)
        steps_per_epoch = (sample_num - 1) // batch_size + 1
        steps_to_valid = steps_per_epoch // valid_cnt_per_epoch + 1

        # Small datasets are uploaded once and stay resident in HBM; large ones (or SATRANS_STREAM_INPUT=1 / model.stream_input
        # = True) stay on the host and are streamed in double-buffered batches (satrans_amd/pipeline.py).
        import os as _os
        stream = getattr(self, "stream_input", None)
        if stream is None:
            stream = _os.environ.get("SATRANS_STREAM_INPUT", "0") == "1" or sample_num * n_cols * 4 > (8 << 30)
        data = labels = None
        # Resident dataset: the columns are uploaded and assembled on the device by a worker thread while this thread runs the
        # callbacks and draws the first epoch's sample order (a seeded CPU randperm, 11 ns per row) - the order is drawn HERE, in
        # this thread, at the point where the reference's DataLoader iterator draws it (after on_epoch_begin): a callback that
        # seeds or reads the global RNG sees what it sees in the reference.  The worker touches no RNG; its exception, if any,
        # is re-raised where the data is first needed.
        upload = None
        if stream:
            packed = np.concatenate(cols, axis=-1)
            if any(np.issubdtype(c.dtype, np.integer) and c.size and int(c.max()) >= (1 << 24) for c in cols) and \
                    any(not np.issubdtype(c.dtype, np.integer) for c in cols):
                packed = np.concatenate([c.astype(np.float64) for c in cols], axis=-1)   # (as _pack)
            host_ids, host_dense = self._host_matrices(packed)
        else:
            import threading
            box = {}

            def _upload():
                try:
                    if str(self.device).startswith("cuda"):
                        torch.cuda.set_device(self.device)
                    box["data"] = self._device_matrix_from_columns(cols)     # whole training set resident in HBM, assembled there
                    box["labels"] = torch.from_numpy(y).to(self.device)
                except BaseException as ex:                                   # noqa: BLE001 - handed to the caller's thread
                    box["error"] = ex
            upload = threading.Thread(target=_upload)
            upload.start()
        self.train()

        # SATRANS_HOST_METRICS=1: per-step train metrics through sklearn on host copies, as the reference does
        from . import device_metrics as DM
        import os
        device_metrics = os.environ.get("SATRANS_HOST_METRICS", "0") != "1" and all(n in DM.BY_NAME for n in self.metrics)
        cbs = CallbackList((callbacks or []) + [self.history])
        cbs.set_model(self)
        cbs.on_train_begin()
        self.stop_training = False
        if verbose:
            print("Train on {0} samples, validate on {1} samples, {2} steps per epoch".format(
                sample_num, len(val_y), steps_per_epoch))

        for epoch in range(initial_epoch, epochs):
            cbs.on_epoch_begin(epoch)
            start_time = time.time()
            train_result: Dict[str, list] = {}
            # [steps, 2] float64 on the device: (log loss, AUC) of every step, one launch each.  Filled here, at the epoch start: the
            # launches that write its rows run on the side stream and must not race the fill
            fused_buf = None
            if verbose > 0 and self.metrics and device_metrics and str(self.device).startswith("cuda"):
                fused_buf = torch.full((steps_per_epoch, 2), float("nan"), dtype=torch.float64, device=self.device)
            # (resident dataset: the order is drawn by a worker thread and handed out head first - the first chunk of steps starts
            #  while the rest of the permutation is still being drawn, sampler.AsyncOrder)
            order = self._epoch_order(sample_num, shuffle, on_host=True, lazy=not stream)
            if upload is not None:
                upload.join()
                upload = None
                if "error" in box:
                    raise box["error"]
                data, labels = box["data"], box["labels"]
            engine.reset_epoch_sums()
            if not stream and torch.is_tensor(data):
                # several ranks, owner form: the epoch's exchange sizes at once (the one consumer that needs the whole order now)
                whole = order.full() if order is not None and engine.plans_owner_counts() else None
                engine.plan_owner_counts(data, whole, batch_size)
            feeder = None
            if stream:
                from .pipeline import HostBatchFeeder
                feeder = iter(HostBatchFeeder(host_ids, y, batch_size, self.device,
                                              order.numpy() if order is not None else None, host_dense))
            iterator = range(steps_per_epoch)
            bar = None
            if verbose == 1:
                from tqdm import tqdm
                bar = tqdm(iterator)
                iterator = bar
            step_num = 0

            # Shuffled epoch over a resident dataset: the rows of `chunk_steps` steps are gathered by ONE index_select (50 MB at the
            # AliCCP shape), a step's batch is a slice of it - two launches per 64 steps on the launch stream instead of two per step
            chunk_steps = 64
            chunk = {"id": -1}

            def resident_batch(step_):
                lo_, hi_ = step_ * batch_size, min(sample_num, (step_ + 1) * batch_size)
                if order is None:
                    return data[lo_:hi_], labels[lo_:hi_]
                c = step_ // chunk_steps
                if chunk["id"] != c:
                    c_lo = c * chunk_steps * batch_size
                    idx = order.rows(c_lo, min(sample_num, c_lo + chunk_steps * batch_size))
                    chunk.update(id=c, lo=c_lo, data=data.index_select(0, idx), labels=labels.index_select(0, idx))
                return chunk["data"][lo_ - chunk["lo"]:hi_ - chunk["lo"]], chunk["labels"][lo_ - chunk["lo"]:hi_ - chunk["lo"]]

            ahead = resident_batch(0) if feeder is None else None      # (resident dataset: batches are cut one step ahead)
            for step in iterator:
                lo, hi = step * batch_size, min(sample_num, (step + 1) * batch_size)
                if feeder is not None:
                    xb, yb = next(feeder)
                    engine.train_step(xb, yb)
                else:
                    xb, yb = ahead
                    ahead = resident_batch(step + 1) if step + 1 < steps_per_epoch else None
                    # the next batch's id matrix as a hint: its ids -> rows, sort and scenario bucketing run on a side stream
                    # underneath this step's tail kernels (engine._prepare_async)
                    engine.train_step(xb, yb, next_X=ahead[0] if ahead is not None and torch.is_tensor(ahead[0]) else None)
                if verbose > 0 and self.metrics and device_metrics:
                    # the reference's per-step train metrics (:330-337) evaluated on the device: no sync, read once per epoch
                    prob = engine.last_prob()
                    if DM.fused_supported(self.metrics, hi - lo, yb, prob):
                        if fused_buf is None:
                            fused_buf = torch.full((steps_per_epoch, 2), float("nan"), dtype=torch.float64, device=prob.device)
                        # (one workgroup, ~57 us: on the stream of the step's tail kernels, beside the touched-row chain - on the
                        #  launch stream it sat between two steps)
                        out_row = fused_buf[step]
                        engine.after_step(lambda: DM.fused_logloss_auc(yb, prob, out_row), yb)
                        for name in self.metrics:
                            train_result.setdefault(name, []).append(fused_buf[step, 1 if name == "auc" else 0])
                    else:
                        for name in self.metrics:
                            train_result.setdefault(name, []).append(DM.BY_NAME[name](yb, prob))
                elif verbose > 0 and self.metrics:
                    # ... or exactly as the reference computes them, on host copies (forces a device sync every step)
                    y_np = yb.cpu().numpy()
                    p_np = engine.last_prob().cpu().numpy().astype("float64")
                    for name, fn in self.metrics.items():
                        train_result.setdefault(name, []).append(fn(y_np, p_np))
                step_num += 1
                if valid_cnt_per_epoch > 1 and step_num % steps_to_valid == 0 and do_validation:
                    res = self.evaluate(val_x, val_y, batch_size)
                    print(f'Step: {step_num}/{steps_per_epoch}, ' +
                          ''.join(" - %s: %.4f" % (k, v) for k, v in res.items()))
                    self.train()
            if bar is not None:
                bar.close()
            if shuffle and epoch + 1 < epochs and not self.stop_training:
                self._speculate_epoch_order(sample_num)      # (host work under the device's queued tail, before the first read-back)
            engine.raise_if_bad_ids()
            bce_sum, reg_sum = engine.epoch_sums()
            epoch_logs = {"loss": (bce_sum + reg_sum) / sample_num}
            for name, result in train_result.items():
                if result and torch.is_tensor(result[0]):
                    result = torch.stack(result).cpu().numpy()           # the epoch's only read of the per-step metrics
                    one_class = bool(name == "auc" and np.isnan(result).any())
                    if name == "auc":
                        # several ranks: every rank must leave the epoch the same way, or the others block in their next
                        # collective - the flag is agreed on before anybody raises
                        from . import parallel as _par
                        if _par.world_size() > 1:
                            one_class = bool(_par.all_reduce_scalars(torch.tensor([float(one_class)], dtype=torch.float64,
                                                                                  device=self.device)).item() > 0)
                    if one_class:
                        # sklearn (the reference's per-step call) raises on such a batch; here it surfaces at the epoch end
                        raise ValueError("Only one class present in y_true of a training batch. ROC AUC score is not "
                                         "defined in that case.")
                epoch_logs[name] = np.sum(result) / steps_per_epoch
            if do_validation:
                for name, result in self.evaluate(val_x, val_y, batch_size).items():
                    epoch_logs["val_" + name] = result
                self.train()
            if verbose > 0:
                msg = "{0}s - loss: {1: .4f}".format(int(time.time() - start_time), epoch_logs["loss"])
                for name in self.metrics:
                    if name in epoch_logs:
                        msg += " - " + name + ": {0: .4f}".format(epoch_logs[name])
                    if do_validation:
                        msg += " - val_" + name + ": {0: .4f}".format(epoch_logs["val_" + name])
                print('Epoch {0}/{1}'.format(epoch + 1, epochs))
                print(msg)
            cbs.on_epoch_end(epoch, epoch_logs)
            if self.stop_training:
                break
        if upload is not None:                                   # (no epoch ran: initial_epoch >= epochs)
            upload.join()
        cbs.on_train_end()
        return self.history

    def _check_ranks_agree(self, sample_num: int, batch_size: int) -> None:
        """Data-parallel runs: every rank must bring the same number of samples and the same batch size, because
        the step's collectives are sized from the local batch (all_gather_into_tensor of B*F_large rows per rank) and
        every rank must take the same number of steps.  A mismatch would hang or mis-size a collective, so it is an
        error here (the ragged LAST batch is fine: it has the same size on every rank)."""
        from . import parallel
        if parallel.world_size() == 1:
            return
        import torch.distributed as dist
        dev = self.device if (str(self.device).startswith("cuda") and dist.get_backend() != "gloo") else "cpu"
        lo = torch.tensor([sample_num, batch_size], dtype=torch.int64, device=dev)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise ValueError(
                f"data-parallel fit(): ranks disagree on (samples, batch_size): min {lo.tolist()} max {hi.tolist()}, "
                f"this rank ({sample_num}, {batch_size}). Give every rank an equally long shard (drop or pad the tail) "
                f"and the same batch_size.")

    def _epoch_order(self, n: int, shuffle: bool, on_host: bool = False, lazy: bool = False):
        """Sample order of one epoch.  With shuffle the permutation is drawn the way
        torch.utils.data.DataLoader(shuffle=True) draws it for the reference (one base-seed draw by the loader
        iterator, one seed draw by RandomSampler, then randperm with that seed), so the same torch seed yields
        the same batches.  The permutation itself comes from satrans_amd/sampler.py: torch.randperm's result bit for bit, drawn
        faster; `lazy`: a sampler.AsyncOrder - a worker thread draws it, its head is usable before its tail exists."""
        if not shuffle:
            return None
        from . import sampler
        torch.empty((), dtype=torch.int64).random_()
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        spec, self._order_spec = getattr(self, "_order_spec", None), None
        ahead = spec[2] if spec is not None and spec[0] == seed and spec[1] == n else None
        # (`ahead`: drawn ahead from the seed this point was going to produce, _speculate_epoch_order)
        if lazy:
            return sampler.AsyncOrder(seed, n, self.device, ready=ahead)
        perm = ahead if ahead is not None else sampler.randperm(seed, n)
        return perm if on_host else perm.to(self.device)

    def _speculate_epoch_order(self, n: int) -> None:
        """The NEXT epoch's permutation, drawn while the device still works off the tail of this epoch's launch queue (the host
        runs ~0.3 ms per step ahead of it and would otherwise sit in the epoch's first read-back).  The permutation is a function
        of the seed that the global generator's next-but-one draw yields: that draw is PEEKED (state saved, two draws, state
        restored - the generator is left exactly as it was), the 11 ns / row randperm runs on a private generator, and
        `_epoch_order` uses the result only if its own, real draws - after the epoch-end and epoch-begin callbacks, where the
        reference's DataLoader draws - produce the same seed.  A callback that reseeds or consumes the generator in between simply
        makes the speculation miss.  (A 17 ms serial bubble per 1.6 M rows and epoch, 9 % of a warm epoch, before.)"""
        state = torch.get_rng_state()
        torch.empty((), dtype=torch.int64).random_()
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        torch.set_rng_state(state)
        from . import sampler
        self._order_spec = (seed, n, sampler.randperm(seed, n))

    def evaluate(self, x, y, batch_size=256):
        """Metric name -> value on (x, y); models/meta_basemodel.py:387-399."""
        pred = self.predict(x, batch_size, y)
        return {name: fn(y, pred) for name, fn in self.metrics.items()}

    def evaluate_domains(self, x, y, batch_size=256, domain_col=None):
        """The test report of reference main.py:353-374 in one call: predict, overall ROC AUC, one AUC per scenario id of
        `domain_col` (default: the model's first scenario column) and the test BCE.  The metrics are evaluated on the device
        (satrans_amd/device_metrics.py, equal to sklearn's), one host read at the end.
        -> {"auc": float, "domain_auc": {id: float}, "loss": float, "pred": float64 [N,1]}"""
        from . import device_metrics as DM
        if domain_col is None:
            domain_col = self.domain_column_list[0]
        pred = self.predict(x, batch_size)
        dev = self.device
        ids = x[domain_col] if isinstance(x, dict) else self._pack(x)[:, self.feature_index[domain_col][0]]
        auc, per, loss = DM.per_domain_auc(torch.as_tensor(np.asarray(y, dtype=np.float64)).to(dev),
                                           torch.from_numpy(pred).to(dev),
                                           torch.as_tensor(np.asarray(ids).astype(np.int64)).to(dev))
        return {"auc": auc, "domain_auc": per, "loss": loss, "pred": pred}

    def predict(self, x, batch_size=256, y=None, domain_ids=None):
        """float64 [N,1] probabilities; models/meta_basemodel.py:401-517 (without the showattn/instattn
        paper-figure branches)."""
        engine = self._require_engine()
        was_training = self.training
        self.eval()
        is_matrix = isinstance(x, np.ndarray) and x.ndim == 2 and not isinstance(x, (dict, list))
        cols = [x] if is_matrix else self._columns(x)
        n_rows, n_cols = cols[0].shape[0], sum(c.shape[1] for c in cols)
        import os as _os
        stream = getattr(self, "stream_input", None)
        if stream is None:
            stream = _os.environ.get("SATRANS_STREAM_INPUT", "0") == "1" or n_rows * n_cols * 4 > (8 << 30)
        out = torch.empty((n_rows, 1), dtype=torch.float32, device=self.device)
        if stream:                                      # host-resident, double-buffered (satrans_amd/pipeline.py)
            from .pipeline import HostBatchFeeder
            packed = x if is_matrix else self._pack(x)
            host_ids, host_dense = self._host_matrices(packed)
            lo = 0
            for xb, _ in HostBatchFeeder(host_ids, None, batch_size, self.device, None, host_dense):
                out[lo:lo + len(xb)] = engine.forward(xb, training=False)
                lo += len(xb)
        else:
            data = self._to_device_matrix(x) if is_matrix else self._device_matrix_from_columns(cols)
            for lo in range(0, data.shape[0], batch_size):
                hi = min(data.shape[0], lo + batch_size)
                out[lo:hi] = engine.forward(data[lo:hi], training=False)
        engine.raise_if_bad_ids()
        if was_training:
            self.train()
        return out.cpu().numpy().astype("float64")

    # ------------------------------------------------------------------------------------------
    def _require_engine(self):
        raise NotImplementedError

    def _flush_engine(self):
        """Postponed optimizer steps land; single-rank entry points (state_dict, get_regularization_loss): never a collective."""
        eng = getattr(self, "_engine", None)
        if eng is not None:
            eng.flush_lazy(sync="local")

    def synchronize(self):
        """Data-parallel runs (owner form): bring this rank's postponed optimizer steps and every rank's copy of the embedding
        tables up to date.  COLLECTIVE: call it on every rank (fit / predict / evaluate do so themselves at their flush points;
        this is for a checkpoint in the middle of an epoch).  One rank: the same as the flush state_dict() performs."""
        eng = getattr(self, "_engine", None)
        if eng is not None:
            eng.flush_lazy()

    def state_dict(self, *args, **kwargs):
        """The tables are brought up to date first (postponed regulariser-only Adam steps, see engine.flush_lazy)."""
        self._flush_engine()
        return super().state_dict(*args, **kwargs)

    def get_regularization_loss(self):
        """sum(l2 * w^2) over the embedding tables (models/meta_basemodel.py:577-593) as a [1] tensor.
        A convenience for callers; the training step computes the same sum inside the optimizer kernels."""
        self._flush_engine()
        total = torch.zeros((1,), device=self.embedding_arena.device)
        if self.l2_reg_embedding > 0:
            total += self.l2_reg_embedding * torch.sum(torch.square(self.embedding_arena.double())).float()
        return total

    def compute_input_dim(self, feature_columns, include_sparse=True, include_dense=True, feature_group=False):
        sparse, dense, varlen = split_columns(feature_columns)
        total = 0
        if include_sparse:
            total += len(sparse + varlen) if feature_group else sum(c.embedding_dim for c in sparse + varlen)
        if include_dense:
            total += sum(c.dimension for c in dense)
        return total
