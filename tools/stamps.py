"""Diagnostic: per-phase cycle shares of the fused backward kernel.  Needs a library built with
SATRANS_EXTRA_FLAGS=-DSATRANS_STAMPS (bash satrans_amd/csrc/build.sh after touching layer_fused.hip).

Read the shares, not the totals: a stamp is an s_memtime plus a global atomic, and since the backward's loads are issued a
phase ahead every `s_waitcnt vmcnt` behind a stamp also waits for that atomic - the instrumented kernel is
~10 % slower than the shipped one and phases that start with a wait (C, D) look longer than they are."""
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
RS = False
buf = (C.c_ulonglong * 32)()
read = lib.satrans_debug_read_stamps
for i in range(2):
    eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B])
torch.cuda.synchronize()
read(buf, 1)
for i in range(2, 4):
    eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B])
torch.cuda.synchronize()
read(buf, 0)
names = ["tile top / dx tail", "A fwd chain", "B attn fwd", "C out block", "D rows", "E cols", "F metanet/proj bwd",
         "rec flush", "prologue"]
# slot k holds the time BEFORE stamp k (stamps sit behind the phase's closing barrier, so a slot includes the wait for the
# slowest wave): 0 = loop top, 1 = phase A, ... 6 = phase F, 7 = record flush, 8 = prologue; 6 backward launches (2 steps x 3 layers)
for role, off in ((("Q-role wave 0", 0), ("K-role wave 4", 16)) if RS else (("wave 0", 0),)):
    vals = [buf[off + i] for i in range(9)]
    tot = sum(vals)
    if tot == 0:
        print(f"{role}: no stamps recorded (is the library built with SATRANS_EXTRA_FLAGS=-DSATRANS_STAMPS?)")
        continue
    print(f"--- {role}: {tot / (6 * 256) / 1e3:.1f} kcycles per workgroup-launch")
    for nme, v in zip(names, vals):
        print(f"{nme:26s} {v / tot * 100:6.2f} %   {v / (6 * 256) / 1e3:9.1f} kcycles per workgroup-launch")
# the forward kernel's slots (three launches per step, two steps; wave 0 of every workgroup)
fnames = ["fwd staging / loop top", "fwd 1 projections + MetaNet", "fwd 2 attention", "fwd 3 out block"]
fvals = [buf[i] for i in range(9, 13)]
ftot = sum(fvals) or 1
print(f"--- forward, wave 0: {ftot / (6 * 256) / 1e3:.1f} kcycles per workgroup-launch")
for nme, v in zip(fnames, fvals):
    print(f"{nme:28s} {v / ftot * 100:6.2f} %   {v / (6 * 256) / 1e3:9.1f} kcycles per workgroup-launch")
