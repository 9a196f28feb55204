"""7,000 pipelined steps in blocks of 256: ms/step, zero fractions and finiteness of the table state per block (a soak run)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
cfg = bench.make_config("aliccp")
bench.CFG = cfg
B, n = 8192, 64
X, y = bench.synth_batches(n * B, 5, cfg=cfg)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
model = bench.build_model("cpu", cfg["lr"], cfg=cfg)
model.to("cuda:0"); model.device = "cuda:0"
model.train()
eng = model._require_engine()
for r in range(28):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(256):
        i = k % (n - 1)
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
    eng.flush_lazy(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 256 * 1e3
    p, m = model.embedding_arena, eng.adam_m
    print(f"steps {eng.adam_t:5d}: {dt:.4f} ms/step  zero p {float((p == 0).float().mean()):.3f} zero m {float((m == 0).float().mean()):.3f} "
          f"finite {bool(torch.isfinite(p).all())} loss_sum {float(eng.loss_sum.item()):.1f}", flush=True)
