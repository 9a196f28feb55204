"""History['loss'] of fit() against the per-step binary_crossentropy metric (an independent kernel) on bench-shaped data."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
cfg = bench.make_config("aliccp")
model = bench.build_model("cpu", cfg["lr"], cfg=cfg)
model.to("cuda:0"); model.device = "cuda:0"
B, n = 8192, 60
X, y = bench.synth_batches(n * B, 777, cfg=cfg)
names = list(cfg["fields"])
x = {f: X[:, i].astype(np.int64) for i, f in enumerate(names)}
for ep in range(3):
    h = model.fit(x=x, y=y, batch_size=B, epochs=1, verbose=1, shuffle=True)
    print({k: float(v[-1]) for k, v in h.history.items()}, "label mean", float(y.mean()), flush=True)
eng = model._require_engine()
print("flush_count", getattr(eng, "flush_count", None), "lazy", eng.lazy, "rolling", eng.rolling_flush)
