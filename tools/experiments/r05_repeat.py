"""Is the step time a function of how long the process has been running?  256 pipelined steps (flushes inside) x 6 repetitions,
then 5 s of idle and 2 more, then a fresh model in the same process and 2 more."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
cfg = bench.make_config("aliccp")
bench.CFG = cfg
B, n = 8192, 64
X, y = bench.synth_batches(n * B, 5, cfg=cfg)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()


def fresh():
    model = bench.build_model("cpu", cfg["lr"], cfg=cfg)
    model.to("cuda:0"); model.device = "cuda:0"
    model.train()
    return model, model._require_engine()


def run(eng, steps):
    for i in range(5):
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
    eng.flush_lazy(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        i = 5 + k % (n - 6)
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
    eng.flush_lazy(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def phases(eng):
    """medians of the recorded-event phases and of the layer kernels' own durations over 16 steps"""
    eng.timers = {}
    eng.untimed_phases = frozenset(bench.LAYER_PHASES)
    eng.lib.satrans_kernel_timing(1)
    for k in range(16):
        i = 5 + k
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
    eng.lib.satrans_kernel_timing(0)
    torch.cuda.synchronize()
    ph = eng.phase_ms()
    ph.update(bench.read_dispatch_ms(eng.lib))
    eng.timers = None
    return " ".join(f"{k}={v * 1e3:.0f}" for k, v in sorted(ph.items()))


model, eng = fresh()
for r in range(6):
    print(f"repetition {r}: {run(eng, 256):.4f} ms/step (adam_t {eng.adam_t})", flush=True)
    if r in (0, 2, 5):
        print("   phases us:", phases(eng), flush=True)
time.sleep(5)
for r in range(2):
    print(f"after 5 s idle {r}: {run(eng, 256):.4f} ms/step", flush=True)
model2, eng2 = fresh()
for r in range(2):
    print(f"fresh model {r}: {run(eng2, 256):.4f} ms/step (adam_t {eng2.adam_t})", flush=True)
