"""Diagnostic: per-phase cycle shares of the fused backward kernel.  Needs a library built with
SATRANS_EXTRA_FLAGS=-DSATRANS_STAMPS (bash satrans_amd/csrc/build.sh after touching layer_fused.hip)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.run(["touch", os.path.join(ROOT, "satrans_amd/csrc/layer_fused_common.h")], check=True)
subprocess.run(["bash", os.path.join(ROOT, "satrans_amd/csrc/build.sh")], check=True,
               env=dict(os.environ, SATRANS_EXTRA_FLAGS="-DSATRANS_STAMPS"), stdout=subprocess.DEVNULL)
import atexit  # noqa: E402


def _restore():
    """Leave the product library behind, not the instrumented one."""
    subprocess.run(["touch", os.path.join(ROOT, "satrans_amd/csrc/layer_fused_common.h")], check=False)
    subprocess.run(["bash", os.path.join(ROOT, "satrans_amd/csrc/build.sh")], check=False, stdout=subprocess.DEVNULL)


atexit.register(_restore)
import torch  # noqa: E402
import bench  # noqa: E402

model = bench.build_model("cpu", 0.005)
model.to("cuda:0"); model.device = "cuda:0"
eng = model._require_engine()
eng.overlap = False
B = 8192
X, y = bench.synth_batches(4 * B, 5)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
model.train()
lib = C.CDLL(os.path.join(ROOT, "satrans_amd/libsatrans_hip.so"))
buf = (C.c_ulonglong * 16)()
for i in range(2):
    eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B])
torch.cuda.synchronize()
lib.satrans_debug_read_stamps(buf, 1)
if os.environ.get("SATRANS_BWD8", "1") != "0":
    lib.satrans_debug_read_stamps8(buf, 1)
for i in range(2, 4):
    eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B])
torch.cuda.synchronize()
lib.satrans_debug_read_stamps(buf, 0)
if os.environ.get("SATRANS_BWD8", "1") != "0":
    lib.satrans_debug_read_stamps8(buf, 0)
    names8 = ["tile top", "A fwd chain", "B attn fwd", "C out block", "D dWo + softmax bwd", "D write-back", "F round 1 compute",
              "F round 1 products", "F round 2 compute", "F round 2 products", "F round 3 stores", "F round 3 products + dx",
              "flush records", "prologue"]
    vals = [buf[i] for i in range(14)]
    tot = sum(vals)
    for n, v in zip(names8, vals):
        print(f"{n:26s} {v / tot * 100:6.2f} %   {v / (6 * 256) / 1e3:9.1f} kcycles per workgroup-launch")
    sys.exit(0)
names = ["stage weights(scenario)", "A fwd chain", "B attn fwd", "C out block", "D rows", "E cols", "F metanet/proj bwd",
         "rec flush", "prologue"]
vals = [buf[i] for i in range(9)]
tot = sum(vals)
# slot k holds the time BEFORE stamp k: 0 = scenario staging/loop top, 1 = phase A, ... 6 = phase F, 7 = record flush, 8 = prologue
for n, v in zip(names, vals):
    print(f"{n:26s} {v / tot * 100:6.2f} %   {v / (6 * 256) / 1e3:9.1f} kcycles per workgroup-launch")
# the forward kernel's slots (three launches per step, two steps)
fnames = ["fwd staging / loop top", "fwd 1 projections + MetaNet", "fwd 2 attention", "fwd 3 out block"]
fvals = [buf[i] for i in range(9, 13)]
ftot = sum(fvals) or 1
for n, v in zip(fnames, fvals):
    print(f"{n:26s} {v / ftot * 100:6.2f} %   {v / (6 * 256) / 1e3:9.1f} kcycles per workgroup-launch")
