"""Sample order of a shuffled epoch: the permutation the reference's loader iterates over (models/meta_basemodel.py:279-280:
DataLoader(shuffle=True) -> RandomSampler -> torch.randperm(n, generator=g), g seeded from the global generator).

torch draws it on the CPU at 11-75 ns per row - 17 ms for the 1.6 M rows of the bench's fit leg, 1.4 s for AliCCP's 42 M rows -
and an epoch cannot take its first step without it.  `satrans_host_randperm` (csrc/host_sampler.hip, a HOST function of the
C-ABI library) is the same Fisher-Yates pass over the same Mersenne-Twister draws with the swap partners prefetched a block ahead,
and it publishes how many leading positions are final while it runs: `AsyncOrder` lets fit() start on the head of the order while
a worker thread draws the rest.  The result is torch's permutation bit for bit - checked against torch.randperm once per process
(`native_ok`); if the check ever fails (another torch algorithm), or n is beyond the 32-bit form, torch.randperm is used as before.
Host logic only: there is no device arithmetic here to fall back from."""
from __future__ import annotations

import ctypes as C
import threading
import time
from typing import Optional

import numpy as np
import torch

_NATIVE: Optional[bool] = None
_LIMIT = 0xFFFFFFFF // 20          # torch.randperm switches algorithms at n >= 2**32 / 20


def _native(seed: int, n: int, out: torch.Tensor, progress=None) -> None:
    from . import native as N
    N.check(N.lib().satrans_host_randperm(C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), n, out.data_ptr(),
                                          progress.ctypes.data if progress is not None else None), "satrans_host_randperm")


def native_ok() -> bool:
    """True when the library's pass reproduces torch.randperm (a 1,000-row and a 70,001-row draw, a seed above 2**32)."""
    global _NATIVE
    if _NATIVE is None:
        try:
            ok = True
            for seed, n in ((2 ** 40 + 12345, 1000), (77, 70001)):
                gen = torch.Generator()
                gen.manual_seed(seed)
                want = torch.randperm(n, generator=gen)
                got = torch.empty(n, dtype=torch.int64)
                _native(seed, n, got)
                ok = ok and bool(torch.equal(want, got))
            _NATIVE = ok
        except Exception:      # noqa: BLE001 - no library on this host: torch draws the order, as before
            _NATIVE = False
    return _NATIVE


def randperm(seed: int, n: int) -> torch.Tensor:
    """torch.randperm(n, generator=Generator().manual_seed(seed)) on the host."""
    if 1 < n < _LIMIT and native_ok():
        out = torch.empty(n, dtype=torch.int64)
        _native(seed, n, out)
        return out
    gen = torch.Generator()
    gen.manual_seed(seed)
    return torch.randperm(n, generator=gen)


class AsyncOrder:
    """The permutation of (seed, n), drawn by a worker thread; `rows(lo, hi)` hands out a slice on `device` as soon as its
    positions are final.  The head (the first request) is copied from the host as soon as it is final; the worker uploads the whole
    order when the pass is done, and later requests are slices of that tensor."""

    def __init__(self, seed: int, n: int, device, ready: Optional[torch.Tensor] = None):
        self.n, self.device = n, device
        self._full = None           # the whole order on the device
        self._event = None
        self._error = None
        self._thread = None
        self._progress = np.zeros(1, dtype=np.int64)
        if ready is not None:       # drawn ahead (basemodel._speculate_epoch_order)
            self.host = ready
            self._progress[0] = n
        elif 1 < n < _LIMIT and native_ok():
            self.host = torch.empty(n, dtype=torch.int64)
            self._thread = threading.Thread(target=self._work, args=(seed,))
            self._thread.start()
        else:
            self.host = randperm(seed, n)
            self._progress[0] = n

    def _work(self, seed):
        try:
            _native(seed, self.n, self.host, self._progress)
            if str(self.device).startswith("cuda"):
                # On this thread's current stream - the device's default one, the stream fit() runs its steps on: a 13 MB copy
                # in between two steps.  NOT on a stream of its own: HIP deals streams onto its hardware queues in creation
                # order, and one more stream created here, ahead of the engine's lazily created ones, left the per-step metrics
                # stream serialised with the launch stream (measured: fit(verbose=1) 0.955 -> 1.17 ms per step, every epoch).
                torch.cuda.set_device(self.device)
                full = self.host.to(self.device)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self._full, self._event = full, ev
        except BaseException as ex:      # noqa: BLE001 - re-raised in the consumer's thread
            self._error = ex
            self._progress[0] = self.n

    def _wait(self, upto: int) -> None:
        while int(self._progress[0]) < upto:
            time.sleep(0.0002)
        if self._error is not None:
            raise self._error

    def full_host(self) -> torch.Tensor:
        self._wait(self.n)
        if self._thread is not None:
            self._thread.join()
        if self._error is not None:
            raise self._error
        return self.host

    def full(self) -> torch.Tensor:
        """The whole order on the device (waits for the pass)."""
        host = self.full_host()
        if self._full is None:
            self._full = host.to(self.device)
        elif self._event is not None:
            torch.cuda.current_stream(self.device).wait_event(self._event)
            self._event = None
        return self._full

    def rows(self, lo: int, hi: int) -> torch.Tensor:
        if self._full is None and self._thread is not None and self._thread.is_alive():
            self._wait(hi)
            if self._full is None:
                return self.host[lo:hi].to(self.device)      # the head, while the worker is still drawing the rest
        return self.full()[lo:hi]
