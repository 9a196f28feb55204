"""How long does the host need to ENQUEUE one training step (no synchronisation) vs the GPU to execute it?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

if os.environ.get("SATRANS_FORCE_EXCHANGE") == "1":      # one rank through RCCL: the owner-form step's host cost
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
model = bench.build_model("cpu", 0.005)
model.to("cuda:0"); model.device = "cuda:0"
eng = model._require_engine()
B = 8192
X, y = bench.synth_batches(60 * B, 5)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
model.train()
if os.environ.get("SATRANS_FORCE_EXCHANGE") == "1":
    eng.plan_owner_counts(Xd, None, B)
for i in range(5):
    eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5, 55):
    eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 50:.3f} ms/step ; wall incl. GPU drain {1e3 * (t2 - t0) / 50:.3f} ms/step")
import cProfile, pstats
if os.environ.get("SATRANS_FORCE_EXCHANGE") == "1":
    eng._prep = None
    eng.plan_owner_counts(Xd[5 * B:], None, B)      # (the plan is positional: the loop below starts again at batch 5)
pr = cProfile.Profile(); pr.enable()
for i in range(5, 25):
    eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], Xd[(i + 1) * B:(i + 2) * B])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
