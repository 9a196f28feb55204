"""Kernel durations and host time of the bf16 evaluation forward (engine.forward at 32,768 samples, stack launch on / off):
   rocprofv3 --kernel-trace --stats -d out -o p -- python3 tools/experiments/r06_bf16_forward_trace.py   (or plain python: host vs device time)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
cfg = bench.make_config("aliccp")
model = bench.build_model("cpu", cfg["lr"], cfg=cfg)
model.to("cuda:0"); model.device = "cuda:0"
eng = model._require_engine()
X, y = bench.synth_batches(4 * 32768, 5, cfg=cfg)
Xd = torch.from_numpy(X).cuda()
model.eval()
model.set_forward_precision("bf16")
nb = 32768
for stack in (True, False):
    eng.bf16_stack = stack
    for _ in range(3):
        eng.forward(Xd[:nb])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for i in range(20):
        eng.forward(Xd[(i % 4) * nb:(i % 4 + 1) * nb])
    e1.record(); t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"stack={stack}: device {e0.elapsed_time(e1) / 20:.4f} ms per forward, host enqueue {t_host / 20 * 1e3:.4f} ms per forward", flush=True)
