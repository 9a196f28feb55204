"""How fast can fit()'s resident dataset get to the device?  19 int64 id columns of 1,638,400 rows (the bench's fit leg):
(a) what basemodel._device_matrix_from_columns does (per column: pageable int64 upload + cast into the fp32 matrix on the device),
(b) columns narrowed to int32 on the host by a thread pool, then (a),  (c) narrowed straight into a pinned staging buffer by the
pool, one async copy per column behind its cast.  python tools/experiments/r06_upload.py"""
import sys, os, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch

N, C = 1638400, 19
rng = np.random.RandomState(0)
cols = [rng.randint(0, 300000, size=(N, 1)).astype(np.int64) for _ in range(C)]
dev = "cuda:0"
torch.zeros(1, device=dev); torch.cuda.synchronize()


def a():
    data = torch.empty(N, C, dtype=torch.float32, device=dev)
    for j, c in enumerate(cols):
        data[:, j:j + 1] = torch.from_numpy(c).to(dev)
    return data


def b(pool):
    data = torch.empty(N, C, dtype=torch.float32, device=dev)
    futs = [pool.submit(lambda c=c: c.astype(np.int32)) for c in cols]
    for j, f in enumerate(futs):
        data[:, j:j + 1] = torch.from_numpy(f.result()).to(dev)
    return data


_pin = {}


def c(pool, parts=4):
    data = torch.empty(N, C, dtype=torch.float32, device=dev)
    if "buf" not in _pin:
        _pin["buf"] = torch.empty(C, N, dtype=torch.int32).pin_memory()
    buf = _pin["buf"]
    bn = buf.numpy()
    step = (N + parts - 1) // parts

    def cast(j, lo):
        np.copyto(bn[j, lo:lo + step], cols[j][lo:lo + step, 0], casting="unsafe")
    futs = [[pool.submit(cast, j, lo) for lo in range(0, N, step)] for j in range(C)]
    for j in range(C):
        for f in futs[j]:
            f.result()
        data[:, j] = buf[j].to(dev, non_blocking=True)
    return data


for threads in (4, 8, 16):
    pool = ThreadPoolExecutor(threads)
    for name, fn in (("a pageable int64", a), ("b pool int32 + pageable", lambda: b(pool)), ("c pool -> pinned int32", lambda: c(pool))):
        ts = []
        for r in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            d = fn(); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ref = a() if name[0] != "a" else d
        assert torch.equal(ref, d)
        print(f"threads {threads:2d}  {name:28s} ms: " + " ".join(f"{t:.1f}" for t in ts), flush=True)
    pool.shutdown()
