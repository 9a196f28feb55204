"""How many elements / lanes / waves are outside the packed range of the replay after 64 ... 960 steps (tables after a flush);
output of round 5: profiles/r05_decay_state_stats.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
lr = cfg["lr"]


def stats(tag):
    eng.flush_lazy(); torch.cuda.synchronize()
    p, m, v = model.embedding_arena, eng.adam_m, eng.adam_v
    a = (lr * m).abs()
    bad_v = (v < 2.0 ** -100) | (v > 2.0 ** 64)
    bad_a = (a < 2.0 ** -80) | (a > 2.0 ** 40)
    lane_bad = (bad_v | bad_a).view(-1, 4).any(dim=1)
    wave_bad = lane_bad[: lane_bad.numel() // 64 * 64].view(-1, 64).any(dim=1)
    print(f"{tag}: out of the packed range: v {bad_v.float().mean().item():.2e} a {bad_a.float().mean().item():.2e} "
          f"lanes {lane_bad.float().mean().item():.3e} waves {wave_bad.float().mean().item():.3f}  |p| median "
          f"{p.abs().median().item():.2e}  |m| median {m.abs().median().item():.2e}  v median {v.median().item():.2e}")


k = 0
for stop in (64, 256, 448, 576, 704, 960):
    while k < stop:
        i = k % (n - 1)
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
        k += 1
    stats(f"after {stop} steps")
