"""Input pipeline: host-resident datasets streamed to the GPU in double-buffered batches.

The reference concatenates the whole dataset into one fp32 tensor and lets a Python DataLoader copy a batch per step
(models/meta_basemodel.py:257-284, 311-312); its data come from HDF5 files read into dicts of numpy columns
(utils.py:22-30, 266-278; written by aliccp_dataset_processing.py:237-242).  `BaseModel.fit` keeps small datasets resident in
HBM (one upload).  This module is the other regime - a full AliCCP epoch (42 M rows x 19 columns) or anything that should not
occupy HBM:

  * `HostBatchFeeder`: the packed [N, C] matrix (+ labels, + dense block) stays on the host (numpy, possibly memory-mapped);
    every batch is gathered into one of two PINNED staging buffers and copied to one of two device buffers on a side stream
    while the previous step computes; integer ids travel as integers (no 2**24 limit), dense features as a float block.
  * `load_npy_columns` / `load_h5_columns`: dict-of-columns loaders with the reference's column naming
    (`<split>/<column>`); the HDF5 side is read by satrans_amd/h5lite.py (no h5py in this image), memory-mapped.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import numpy as np
import torch

from .inputs import PackedInput


def load_npy_columns(directory: str, columns: Iterable[str], split: Optional[str] = None, mmap: bool = True) -> Dict[str, np.ndarray]:
    """{column: array} from `<directory>/[<split>/]<column>.npy` (memory-mapped by default: nothing is read until a batch is
    gathered)."""
    base = os.path.join(directory, split) if split else directory
    return {c: np.load(os.path.join(base, f"{c}.npy"), mmap_mode="r" if mmap else None) for c in columns}


def load_h5_columns(path: str, group: Optional[str], columns: Optional[Iterable[str]] = None, mmap: bool = True) -> Dict[str, np.ndarray]:
    """{column: array} from the datasets `<group>/<column>` of an HDF5 file - the layout of the reference's `alicpp.h5`
    (`ctr_train/<col>`, `ctr_test/<col>`; utils.py:266-278) or, with group None, of `alimama.h5` (root-level datasets;
    utils.py:22-30).  Read by satrans_amd/h5lite.py, a parser of what `h5py.File(path, 'w')` + `f[name] = array` write
    (no h5py needed; memory-mapped by default); files outside that subset (chunked / compressed datasets, libver='latest')
    fall back to h5py when it is installed."""
    from . import h5lite
    try:
        return h5lite.read_h5_columns(path, group, columns, mmap=mmap)
    except NotImplementedError as sub:
        try:
            import h5py
        except ImportError:
            raise sub
        with h5py.File(path, "r") as f:
            g = f[group] if group else f
            return {c: g[c][:] for c in (columns if columns is not None else g.keys())}


class HostBatchFeeder:
    """Batches of a host-resident dataset on the device, double-buffered.

    ids    [N, C] host matrix (float32 in the reference's layout, or an integer dtype), labels [N] or None, dense [N, nd] or None.
    Iterating yields (X, y) for consecutive batches of `order` (a host index array; None = in order); X is a device tensor, or a
    `PackedInput` when a dense block is given.  While batch i is being consumed, batch i+1 is already being staged: the host
    gather fills a pinned buffer and the copy runs on its own stream; the consumer's stream waits on the copy's event, and the
    copy of batch i+2 waits until the work issued for batch i has finished with that device buffer."""

    def __init__(self, ids: np.ndarray, labels: Optional[np.ndarray], batch_size: int, device, order: Optional[np.ndarray] = None,
                 dense: Optional[np.ndarray] = None):
        self.ids, self.labels, self.dense = ids, labels, dense
        self.n = ids.shape[0] if order is None else len(order)
        self.order = order
        self.B = int(batch_size)
        self.dev = torch.device(device)
        self.cuda = self.dev.type == "cuda"
        self.steps = (self.n - 1) // self.B + 1 if self.n else 0
        dt_ids = torch.from_numpy(np.empty(0, dtype=ids.dtype)).dtype
        pin = dict(pin_memory=True) if self.cuda else {}
        self._host = [dict(ids=torch.empty(self.B, ids.shape[1], dtype=dt_ids, **pin),
                           y=torch.empty(self.B, dtype=torch.float32, **pin) if labels is not None else None,
                           dense=torch.empty(self.B, dense.shape[1], dtype=torch.float32, **pin) if dense is not None else None)
                      for _ in range(2)]
        self._dev = [dict(ids=torch.empty(self.B, ids.shape[1], dtype=dt_ids, device=self.dev),
                          y=torch.empty(self.B, dtype=torch.float32, device=self.dev) if labels is not None else None,
                          dense=torch.empty(self.B, dense.shape[1], dtype=torch.float32, device=self.dev) if dense is not None else None)
                     for _ in range(2)]
        self._copy = torch.cuda.Stream(self.dev) if self.cuda else None
        self._ready = [None, None]        # copy finished
        self._free = [None, None]         # consumer finished with the device buffer

    def __len__(self):
        return self.steps

    def _stage(self, step: int):
        k = step & 1
        lo, hi = step * self.B, min(self.n, (step + 1) * self.B)
        nb = hi - lo
        idx = slice(lo, hi) if self.order is None else self.order[lo:hi]
        h = self._host[k]
        if self.cuda and self._ready[k] is not None:
            self._ready[k].synchronize()            # the previous copy out of this pinned buffer has left the host
        if self.order is not None:
            np.take(self.ids, idx, axis=0, out=h["ids"].numpy()[:nb])
        else:
            np.copyto(h["ids"].numpy()[:nb], self.ids[idx])
        if self.labels is not None:
            np.copyto(h["y"].numpy()[:nb], np.asarray(self.labels[idx], dtype=np.float32))
        if self.dense is not None:
            np.copyto(h["dense"].numpy()[:nb], np.asarray(self.dense[idx], dtype=np.float32))
        d = self._dev[k]
        if not self.cuda:
            for key in ("ids", "y", "dense"):
                if d[key] is not None:
                    d[key][:nb].copy_(h[key][:nb])
            return nb
        if self._free[k] is not None:
            self._copy.wait_event(self._free[k])    # the step that used this device buffer two batches ago is done
        with torch.cuda.stream(self._copy):
            for key in ("ids", "y", "dense"):
                if d[key] is not None:
                    d[key][:nb].copy_(h[key][:nb], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy)
        self._ready[k] = ev
        return nb

    def __iter__(self):
        if self.steps == 0:
            return
        nb_next = self._stage(0)
        for step in range(self.steps):
            k, nb = step & 1, nb_next
            if step + 1 < self.steps:
                nb_next = self._stage(step + 1)      # overlaps the consumer's work on batch `step`
            d = self._dev[k]
            if self.cuda:
                torch.cuda.current_stream(self.dev).wait_event(self._ready[k])
            X = d["ids"][:nb]
            if d["dense"] is not None:
                X = PackedInput(X, d["dense"][:nb])
            yield X, (d["y"][:nb] if d["y"] is not None else None)
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.dev))
                self._free[k] = ev
