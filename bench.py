#!/usr/bin/env python3
"""Benchmark of the SATrans training step on MI355X (BASELINE.json metric: training samples/sec, AliCCP-shaped
input, embedding_dim 32, 3 layers, 4 heads, meta_mode QK).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full training step of the reference's `fit` loop (models/meta_basemodel.py:310-328) on one batch
per GPU: gather -> 3 layers -> head -> BCE(sum) -> backward -> dense-semantics Adam + L2 over ALL 6.57 M embedding
rows (what the reference's dense `torch.optim.Adam` + `get_regularization_loss` do every step) -> Adam on the
other parameters.  Dropout is on (p = 0.1, four sites per layer).  Inputs are synthetic, AliCCP-shaped
 and already resident in HBM when the timed region starts.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALICCP_FIELDS = ['101', '121', '122', '124', '125', '126', '127', '128', '129', '205', '206', '207', '210', '216',
                 '508', '509', '702', '853', '301']                                # reference main.py:99-101
ALICCP_MAX = {'101': 444861, '121': 97, '122': 13, '124': 2, '125': 7, '126': 3, '127': 3, '128': 2, '129': 4,
              '205': 4348615, '206': 8993, '207': 695124, '210': 99606, '216': 234880, '508': 8185, '509': 472354,
              '702': 167813, '853': 91358, '301': 3}                               # reference main.py:124-127
ALIMAMA_FIELDS = ['user_id', 'adgroup_id', 'pid', 'cms_segid', 'cms_group_id', 'final_gender_code', 'age_level',
                  'pvalue_level', 'shopping_level', 'occupation', 'new_user_class_level', 'cate_id', 'campaign_id', 'customer',
                  'brand']                                                           # reference main.py:143-145
# column maxima: not in the reference tree; public-dataset magnitudes (labelled assumptions)
ALIMAMA_MAX = {'user_id': 1141729, 'adgroup_id': 846811, 'pid': 1, 'cms_segid': 97, 'cms_group_id': 13,
               'final_gender_code': 2, 'age_level': 7, 'pvalue_level': 4, 'shopping_level': 3, 'occupation': 2,
               'new_user_class_level': 5, 'cate_id': 12977, 'campaign_id': 423436, 'customer': 255875, 'brand': 461497}
C5_FIELDS = [f'f{i}' for i in range(63)] + ['dom']


def make_config(name, table_rows=None):
    """The BASELINE.json configs this file can time: aliccp = configs[1] (the headline), alimama = configs[3], c5 = configs[4]."""
    if name == "aliccp":
        return dict(name=name, fields=ALICCP_FIELDS, maxima=ALICCP_MAX, dense=[], domain='301', dom_lo=1, n_domains=3, D=32,
                    H=4, L=3, units=(64, 32), flag='sota', lr=0.005, int_ids=False,
                    label="BASELINE configs[1]: AliCCP-shaped SATrans training step, 19 fields")
    if name == "alimama":
        return dict(name=name, fields=ALIMAMA_FIELDS, maxima=ALIMAMA_MAX, dense=['price'], domain='shopping_level', dom_lo=0,
                    n_domains=3, D=32, H=4, L=3, units=(64, 32), flag='sota-pos', lr=0.001, int_ids=False,
                    label="BASELINE configs[3]: Alimama-shaped SATrans training step, 15 sparse + 1 dense field, flag sota-pos "
                          "(column maxima are public-dataset magnitudes, an assumption)")
    if name == "c5":
        total = int(table_rows or 100_000_000)                  # configs[4]: a 100 M-row embedding table
        per = max(8, total // 63)
        maxima = {f: per - 2 for f in C5_FIELDS[:-1]}
        maxima['dom'] = 3
        return dict(name=name, fields=C5_FIELDS, maxima=maxima, dense=[], domain='dom', dom_lo=1, n_domains=3, D=64, H=4, L=6,
                    units=(128, 64), flag='sota', lr=0.005, int_ids=True,
                    scaled=total < 100_000_000,
                    label="BASELINE configs[4]: synthetic stress, 64 int64 fields, embedding_dim 64, 6 layers, MetaNet hidden 128" +
                          (f" - SCALED: {per * 63 + 5:,} table rows instead of the 100 M configs[4] names" if total < 100_000_000 else
                           " (25.6 GB of tables + 51 GB of Adam moments resident in HBM)"))
    raise ValueError(name)


CFG = make_config("aliccp")

# profiles/: counters of the shipped kernel sources (tools/pmc_passes.sh + pmc_summary.py; configs[4]: tools/pmc_c5.sh +
# pmc_c5_summary.py - there a "launch" is the chain of launches that makes one layer's forward / backward)
PMC_SUMMARIES = {"aliccp": "r06_pmc_summary.json", "alimama": "r06_alimama_pmc_summary.json", "c5": "r06_c5_pmc_summary.json",
                 "aliccp:sota-gate": "r06_gate_pmc_summary.json"}      # key: config, or config:flag for a non-default flag
PMC_SUMMARY = PMC_SUMMARIES["aliccp"]
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured streaming-copy rate
FP32_PEAK_TFLOPS = 157.3       # dense fp32 (vector = f32-input MFMA) peak
BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF figure includes 2:1 sparsity)


def synth_batches(n_rows, seed, ids="uniform", cfg=None):
    """Shaped id matrix of a config.  `uniform`: every id of a field equally likely (the HBM worst case, the bench default);
    `skewed`: log-uniform ranks, P(id) ~ 1/(id+1), i.e. Zipf with exponent 1 (real CTR ids are skewed: a few hot rows)."""
    cfg = cfg or CFG
    rng = np.random.RandomState(seed)
    cols = []
    for f in cfg["fields"]:
        mx = cfg["maxima"][f]
        lo = cfg["dom_lo"] if f == cfg["domain"] else 0                              # AliCCP scenario ids start at 1 (main.py:112-114)
        if ids == "skewed" and f != cfg["domain"]:
            col = np.minimum((np.exp(rng.uniform(0.0, np.log(mx + 1.0), size=n_rows)) - 1.0).astype(np.int64), mx)
        else:
            col = rng.randint(lo, mx + 1, size=n_rows)
        cols.append(col)
    if cfg["int_ids"]:
        X = np.stack(cols, axis=1).astype(np.int64)                                # integer ids (vocabularies beyond fp32's 2**24)
    else:
        cols += [rng.rand(n_rows) for _ in cfg["dense"]]                           # MinMax-scaled price (main.py:156-157)
        X = np.stack(cols, axis=1).astype(np.float32)                              # ids travel as fp32 (meta_basemodel.py:311)
    y = (rng.rand(n_rows) < 0.04).astype(np.float32)                               # assumed CTR level
    return X, y


def build_model(device, lr, flag=None, cfg=None):
    from satrans_amd import DenseFeat, SATrans, SparseFeat
    cfg = cfg or CFG
    cols = [SparseFeat(f, vocabulary_size=cfg["maxima"][f] + 2, embedding_dim=cfg["D"]) for f in cfg["fields"]] + \
           [DenseFeat(f, 1) for f in cfg["dense"]]
    model = SATrans(cols, cols, [cfg["domain"]], [cfg["n_domains"]], att_layer_num=0, domain_att_layer_num=cfg["L"],
                    att_head_num=cfg["H"], use_linear=False, use_dnn=False, meta_mode='QK', meta_dnn_hidden_units=cfg["units"],
                    seed='1021', device=device, flag=flag or cfg["flag"])
    model.compile(torch.optim.Adam(model.parameters(), lr=lr), "binary_crossentropy",
                  metrics=["binary_crossentropy", "auc"])                          # main.py:343
    return model


def oracle_spec(flag=None, cfg=None):
    from oracle.satrans_oracle import PathSpec
    cfg = cfg or CFG
    nf = len(cfg["fields"])
    return PathSpec(sparse=[(f, i) for i, f in enumerate(cfg["fields"])], dense=[(nf + i, nf + i + 1) for i in range(len(cfg["dense"]))],
                    domain_cols=[cfg["fields"].index(cfg["domain"])], embedding_dim=cfg["D"], head_num=cfg["H"],
                    layer_num=cfg["L"], flag=flag or cfg["flag"], meta_mode='QK', meta_units=[cfg["D"]] + list(cfg["units"]))


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(state, X, y, batch, lr, flag='sota', timed=10, warm=2, timed_verbose=5):
    """The reference's training step restated op for op (oracle/satrans_oracle.py: per-sample generated weights, torch CPU
    dropout, dense L2 over all rows, dense torch.optim.Adam), timed on this box's host cores with the protocol of SURVEY
    §8d / BASELINE.md §3: thread count chosen by a sweep that takes the MEDIAN of three steps per candidate (a one-step sweep
    picked 8 threads on one box and 32 on another of the same kind: a 35 % spread in the stated baseline), `warm`
    untimed steps, MEDIAN of `timed` steps; then a second leg with the per-step sklearn log_loss / roc_auc_score of
    `fit(verbose=1)` (what reference main.py does)."""
    from oracle import satrans_oracle as O
    from sklearn.metrics import log_loss, roc_auc_score
    ncpu = os.cpu_count()
    tr = O.OracleTrainer(state, oracle_spec(flag), lr=lr)
    drop = O.Dropper("torch", 0.1)
    cursor = [0]

    def one(verbose=False):
        s = cursor[0]
        cursor[0] += 1
        xb = torch.from_numpy(X[s * batch:(s + 1) * batch])
        yb = torch.from_numpy(y[s * batch:(s + 1) * batch])
        t0 = time.perf_counter()
        prob = tr.step(xb, yb, drop, return_prob=verbose)
        if verbose:                                                                # meta_basemodel.py:330-337
            p64 = prob.numpy().astype("float64")
            log_loss(yb.numpy(), p64, labels=[0, 1])
            if 0 < float(yb.sum()) < len(yb):
                roc_auc_score(yb.numpy(), p64)
        return time.perf_counter() - t0

    torch.set_num_threads(min(ncpu, 32))
    one()                                                                          # allocates Adam state
    # candidates: 8 ... 64 threads (the step is bound by the table sweeps of dense Adam + L2, i.e. by memory bandwidth: on a
    # 256-core host the all-cores candidate ran 69.8 s per step against 1.8 s at 16 threads and was 209 of the bench's 278 s,
    # VERDICT r05); a candidate whose FIRST step is already twice the best median so far is abandoned after that step
    sweep, abandoned = {}, {}
    for th in sorted({t for t in (8, 16, 32, 64, min(ncpu, 64)) if t <= ncpu}):
        torch.set_num_threads(th)
        first = one()
        if sweep and first > 2.0 * min(sweep.values()):
            abandoned[th] = first
            continue
        sweep[th] = sorted([first, one(), one()])[1]
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    for _ in range(max(0, warm - 1)):
        one()
    t_plain = sorted(one() for _ in range(timed))
    t_verb = sorted(one(True) for _ in range(timed_verbose))
    med = lambda v: v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    return dict(value=batch / med(t_plain), value_verbose1=batch / med(t_verb), threads=best, timed=timed,
                timed_verbose=timed_verbose, sweep_s_per_step={str(k): round(v, 3) for k, v in sweep.items()},
                sweep_abandoned_after_one_step={str(k): round(v, 3) for k, v in abandoned.items()}, steps_used=cursor[0])


def gather_microbench(eng, Xd, B, F, D, launches=48):
    """K1 alone (ids -> [B,F,D] rows + the arena row of every id): `launches` back-to-back launches between two HIP events
    on the launch stream, every launch on a different batch of ids (the 841 MB table does not fit the 256 MiB Infinity
    Cache and uniform ids never repeat a row soon), algorithmic bytes F*(D*4+4) read + F*D*4 written per sample."""
    import ctypes as C
    from satrans_amd import native as N
    out = {}
    stream = torch.cuda.current_stream().cuda_stream
    for nb in (B, 4 * B):
        n_batches = Xd.shape[0] // nb
        if n_batches < 2:
            continue
        dst = torch.empty(nb, F, D, dtype=torch.float32, device=Xd.device)
        rows = torch.empty(nb, F, dtype=torch.int32, device=Xd.device)

        def launch(i):
            xb = Xd[(i % n_batches) * nb:(i % n_batches + 1) * nb]
            N.check(eng.lib.satrans_gather_fwd(eng.m.embedding_arena.data_ptr(), eng.row_span.data_ptr(), eng.cols.data_ptr(),
                                               xb.data_ptr(), N.id_dtype_of(xb), xb.stride(0), nb, F, D, dst.data_ptr(),
                                               rows.data_ptr(), eng.status.data_ptr(), stream), "satrans_gather_fwd")
        for i in range(4):
            launch(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(launches):
            launch(i)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / launches
        gbs = nb * F * (2 * D * 4 + 4) / 1e9 / (ms / 1e3)
        rd = nb * F * (D * 4 + 4) / 1e9 / (ms / 1e3)           # READ bytes only: F * (D * 4 + 4) = 2,508 B per sample
        # the read side alone, as the first layer sees it with the gather fused in: rows by row number, nothing written
        sink = torch.empty(int(eng.lib.satrans_gather_read_probe_floats()), dtype=torch.float32, device=Xd.device)
        # a different set of uniformly random arena rows for EVERY launch (48 launches x nb*F rows x 128 B = several GB: no
        # row is served from the 256 MiB Infinity Cache because an earlier launch touched it)
        total = eng.m.embedding_arena.shape[0]
        row_sets = torch.randint(0, total, (launches + 4, nb * F), dtype=torch.int32, device=Xd.device)

        def probe(i):
            N.check(eng.lib.satrans_gather_read_probe(eng.m.embedding_arena.data_ptr(), row_sets[i % row_sets.shape[0]].data_ptr(),
                                                      nb * F, D, sink.data_ptr(), stream), "satrans_gather_read_probe")
        for i in range(4):
            probe(launches + i)
        e0.record()
        for i in range(launches):
            probe(i)
        e1.record()
        torch.cuda.synchronize()
        ms_r = e0.elapsed_time(e1) / launches
        rd_only = nb * F * (D * 4 + 4) / 1e9 / (ms_r / 1e3)
        out[f"batch_{nb}"] = {"read_only_ms_per_launch": round(ms_r, 4), "read_only_GBps": round(rd_only, 1),
                              "read_only_frac_of_peak": round(rd_only / HBM_PEAK_GBS, 4),
                              "ms_per_launch": round(ms, 4), "read_GBps": round(rd, 1), "read_frac_of_peak": round(rd / HBM_PEAK_GBS, 4),
                              "read_plus_write_GBps": round(gbs, 1), "read_plus_write_frac": round(gbs / HBM_PEAK_GBS, 4),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "distinct_id_batches": n_batches}
    return out


LAYER_PHASES = ("layer_fwd_gather", "layer_fwd", "layer_bwd_head", "layer_bwd")      # one fused kernel each (fused path)


def read_dispatch_ms(lib):
    """Median duration of the fused layer kernels' dispatches since the last read (satrans_kernel_timing: HIP events that the
    dispatch packet itself signals with its begin / end timestamps - the kernel alone, on the stream it runs on): phase name -> ms.
    A step's launches arrive in order [layer 0 forward, other forwards ..., last layer + head, backward L-2 .. 0]."""
    import ctypes as C
    kinds, ms = (C.c_int * 512)(), (C.c_float * 512)()
    n = lib.satrans_kernel_timing_read(kinds, ms, 512)
    acc, prev = {}, None
    for i in range(max(0, n)):
        k = kinds[i]
        name = {1: "layer_bwd", 2: "layer_bwd_head"}.get(k) or ("layer_fwd_gather" if prev != 0 else "layer_fwd")
        acc.setdefault(name, []).append(ms[i])
        prev = k
    med = lambda v: sorted(v)[len(v) // 2] if len(v) % 2 else 0.5 * (sorted(v)[len(v) // 2 - 1] + sorted(v)[len(v) // 2])
    return {k: med(v) for k, v in acc.items()}          # (median, as the recorded-event phases: engine.phase_ms)


OTHER_CONFIGS = (          # name, extra arguments, time limit in seconds
    ("configs[3] alimama sota-pos", ["--config", "alimama"], 240),
    ("configs[1] flag sota-gate", ["--flag", "sota-gate"], 240),
    ("configs[1] skewed ids", ["--ids", "skewed"], 240),
    ("configs[4] c5 at 100 M rows", ["--config", "c5", "--with-gather"], 600),
)


def other_configs():
    """`python bench.py <variant> --train-only --steps 20 --warmup 5` as a CHILD process per variant (started, never exec'ed; this
    process keeps its GPU context and waits), each line reduced to {metric, value, ms_per_step, roofline {kernel, bound, frac,
    launch_ms}, kernels ms per launch, gather (configs[4])}.  A variant that fails or overruns its limit is reported as such."""
    import subprocess
    res = {}
    for name, extra, limit in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "20", "--warmup", "5", "--train-only",
               "--no-other-configs"] + extra
        t0 = time.time()
        try:
            cp = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit, text=True)
            line = [l for l in cp.stdout.splitlines() if l.startswith("{")]
            if cp.returncode != 0 or not line:
                res[name] = {"error": f"rc {cp.returncode}", "stderr_tail": cp.stderr[-400:]}
                continue
            d = json.loads(line[-1])
            r = d.get("roofline") or {}
            rec = {"metric": d["metric"], "workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"],
                   "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
                   "roofline": {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launch_ms",
                                                       "launches_per_step", "traffic", "traffic_source")},
                   "kernels_ms_per_launch": {k: v["ms_per_launch"] for k, v in (d.get("kernels") or {}).items()},
                   "wall_s": round(time.time() - t0, 1), "cmd": " ".join(["python", "bench.py"] + cmd[2:])}
            if d.get("gather"):
                rec["gather"] = d["gather"]
            res[name] = rec
        except subprocess.TimeoutExpired:
            res[name] = {"error": f"no line within {limit} s"}
        except Exception as ex:                                                     # never take the headline line down
            res[name] = {"error": repr(ex)[:300]}
    return res


def launch_ranks(n: int) -> int:
    """Start `n` ranks of this script on the GPUs of this node (one process per GPU, RCCL) and wait for them."""
    import socket
    import subprocess
    share = os.environ.get("SATRANS_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()                       # (counts devices without initialising the GPU in this process)
    if have < n and not share:
        print(f"[bench] --gpus {n} but this node shows {have} GPU(s): refusing to report a {have}-GPU number as a {n}-GPU one "
              f"(SATRANS_BENCH_SHARE_GPU=1 runs the {n}-rank step on one GPU over gloo as a functional check)", file=sys.stderr)
        return 2
    with socket.socket() as sock:                          # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"[bench] launching {n} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8192, help="samples per GPU per step (reference main.py:80)")
    ap.add_argument("--lr", type=float, default=0.005)
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed CPU-baseline steps, median reported (0 = skip)")
    ap.add_argument("--train-only", action="store_true",
                    help="only the timed training steps (profiling runs: every kernel row is then one configuration)")
    ap.add_argument("--no-phase-timing", action="store_true")
    ap.add_argument("--config", choices=["aliccp", "alimama", "c5"], default="aliccp",
                    help="aliccp = BASELINE configs[1] (the headline metric), alimama = configs[3] (15 sparse + 1 dense, sota-pos), "
                         "c5 = configs[4] (64 int64 fields, embedding_dim 64, 6 layers)")
    ap.add_argument("--table-rows", type=int, default=None,
                    help="c5: total embedding rows (default: the 100 M configs[4] names; smaller values are labelled SCALED)")
    ap.add_argument("--sustained-steps", type=int, default=None,
                    help="steps of the sustained leg (default 2000 over 200 distinct batches; c5: 120 over 30; 0 = skip)")
    ap.add_argument("--fit-batches", type=int, default=None,
                    help="batches of the fit()-level leg (default 200; c5: 0 = skipped; 0 = skip)")
    ap.add_argument("--flag", default=None, help="SATrans flag (reference main.py --flag); 'sota-pos' = the positional variant")
    ap.add_argument("--ids", choices=["uniform", "skewed"], default="uniform",
                    help="id distribution of the synthetic batches (uniform = HBM worst case, the reported configuration)")
    ap.add_argument("--with-gather", action="store_true", help="with --train-only: also run the gather probes (the c5 line of other_configs)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` object (configs[3], gate, skewed ids, configs[4]: one child process each)")
    args = ap.parse_args()
    import satrans_amd  # noqa: F401  (before the first GPU call: its import sets the HIP runtime's stream-queue default)
    global CFG
    CFG = make_config(args.config, args.table_rows)
    if args.flag is None:
        args.flag = CFG["flag"]
    if args.config != "aliccp" and args.lr == 0.005:
        args.lr = CFG["lr"]

    # ---- `python bench.py --gpus N` with N > 1 and no launcher around it: this process - which has made no GPU call - starts
    #      the N ranks itself (torch.distributed.run as a CHILD process; nothing is exec'ed), relays rank 0's one JSON line and
    #      exits with the children's status.  Under `python -m torch.distributed.run ... bench.py --gpus N` WORLD_SIZE is set
    #      and this branch is not taken. ------------------------------------------------------------------------------------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    # C libraries write to file descriptor 1 behind Python's back (RCCL prints a five-line version banner there): keep the real
    # stdout for the ONE JSON line and send everything else to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_exchange = os.environ.get("SATRANS_FORCE_EXCHANGE", "0") == "1"
    import torch.distributed as dist
    if world == 1 and force_exchange:
        # diagnostic: a ONE-rank nccl group + the step's multi-rank branch (every collective an identity through RCCL): what
        # the exchange code path itself costs on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # SATRANS_BENCH_SHARE_GPU=1 (diagnostic): all ranks on cuda:0 over gloo, to exercise the multi-rank step on a
        # one-GPU box; its numbers are not a scaling measurement
        if os.environ.get("SATRANS_BENCH_SHARE_GPU") == "1":
            local_rank = 0
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if args.gpus != world:
        # the line's n_gpus must be what was asked for: a launcher that started another number of ranks is a usage error
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` "
                         f"(it launches its own ranks) or with --nproc-per-node {args.gpus}")
    if world > 1:
        assert dist.get_world_size() == args.gpus
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    B, K, W = args.batch, args.steps, args.warmup
    t_build = time.time()
    model = build_model("cpu", args.lr, args.flag)                                 # seeded init on CPU, as the reference
    do_cpu = world == 1 and args.cpu_steps > 0 and not args.train_only
    # CPU leg of configs[4]: the oracle's dense step needs ~5 table-sized host buffers (parameters, gradient, two Adam moments,
    # regulariser graph) - 128 GB at 100 M rows.  Its bounded sample therefore runs the same step on 2 M-row tables (said so in
    # `cpu_baseline.sample`); its per-sample cost is dominated by the table sweeps, so this FLATTERS the CPU.
    cpu_cfg = make_config("c5", 2_000_000) if (do_cpu and args.config == "c5" and CFG["maxima"]["f0"] > 40_000) else None
    cpu_src = build_model("cpu", args.lr, args.flag, cfg=cpu_cfg) if cpu_cfg else model
    state_cpu = {k: v.detach().clone() for k, v in cpu_src.state_dict().items()} if do_cpu else None
    if do_cpu:                                                                     # keep the reference's aliasing
        sd = cpu_src.state_dict()
        by_ptr = {}
        for k, v in sd.items():
            state_cpu[k] = by_ptr.setdefault(v.data_ptr(), state_cpu[k])
    n_cpu_rows = cpu_src.embedding_arena.shape[0]
    del cpu_src
    model.to(device)
    model.device = device
    eng = model._require_engine()
    if rank == 0:
        print(f"[bench] model built in {time.time() - t_build:.1f}s; tables {model.embedding_arena.numel() * 4 / 1e6:.0f} MB",
              file=sys.stderr)

    X, y = synth_batches((K + W) * B, seed=100 + rank, ids=args.ids)
    Xd, yd = torch.from_numpy(X).to(device), torch.from_numpy(y).to(device)
    model.train()

    n_res = K + W               # resident batches; a step names the batch that follows it, as `fit` does (engine._prepare_async)

    def step(i):
        j = (i + 1) % n_res
        eng.train_step(Xd[i * B:(i + 1) * B], yd[i * B:(i + 1) * B], next_X=Xd[j * B:(j + 1) * B])

    if world > 1 or force_exchange:
        # several ranks, owner form: the exchange sizes of the W + K resident batches in one pass (what `fit` does per epoch),
        # instead of one count read-back per step
        eng.plan_owner_counts(Xd, None, B)
    for i in range(W):
        step(i)
    # the postponed row updates of the WARM-UP steps belong to the warm-up: bring every row up to date before the clock starts, so
    # that the flush inside the timed region replays exactly the K timed steps (it used to pay for W + K)
    eng.flush_lazy()
    # per-phase HIP events are recorded on every 4th step of the timed region: ~40 event records per step cost
    # ~0.4 ms/step (measured 2.13 vs 1.69 ms/step), which would distort the very number being reported
    timers = {} if not args.no_phase_timing else None
    if not bool(eng._ws.get(B, {}).get("generic")):
        eng.untimed_phases = frozenset(LAYER_PHASES)      # their durations come from the dispatches themselves (read_dispatch_ms)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    flushes_before = getattr(eng, "flush_count", 0)
    t0 = time.perf_counter()
    for i in range(W, W + K):
        sampled = timers is not None and (i - W) % 4 == 0
        eng.timers = timers if sampled else None
        eng.lib.satrans_kernel_timing(1 if sampled else 0)      # the fused layer kernels' own durations on the sampled steps
        step(i)
    eng.lib.satrans_kernel_timing(0)
    eng.timers = timers
    # lazy-exact Adam: every postponed row update of the K steps is paid inside the timed region.  (Several ranks, owner form:
    # each rank flushes the rows it owns - that IS the work of the K steps; bringing the replicas together again, 2.5 GB of
    # slice broadcasts that an epoch pays once, happens after the clock stops.)
    eng.flush_lazy(sync=False)
    n_flush_timed = getattr(eng, "flush_count", 0) - flushes_before
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _t = eng.timers
    eng.timers = None
    eng.flush_lazy()          # (replica synchronisation of the owner form, outside the timed region and its phase timers)
    eng.timers = _t
    per_rank_ms = [elapsed / K * 1e3]
    if world > 1:
        cdev = "cpu" if dist.get_backend() == "gloo" else device
        mine = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        every = torch.zeros(world, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(every, mine)
        per_rank_ms = [float(v) / K * 1e3 for v in every.cpu()]
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        elapsed = float(mine.item())
    from satrans_amd import parallel as _par
    collectives = {k: {"calls_per_step": v["calls"] / (K + W), "bytes_sent_per_step": v["bytes_in"] / (K + W),
                       "bytes_received_per_step": v["bytes_out"] / (K + W)} for k, v in _par.STATS.items()
                   if not k.startswith("broadcast_")}
    # the owner form's replica synchronisation is not a per-step exchange: totals of the run (twice here: before and after the
    # timed region), outside the timed region
    for k, v in _par.STATS.items():
        if k.startswith("broadcast_"):
            collectives[k] = {"calls_total": v["calls"], "bytes_sent_total": v["bytes_in"], "bytes_received_total": v["bytes_out"],
                              "note": "replica synchronisation at flush points, outside the timed region"}
    eng.raise_if_bad_ids()
    phases = eng.phase_ms() if eng.timers is not None else {}
    dispatch = read_dispatch_ms(eng.lib) if eng.timers is not None else {}
    eng.timers = None

    # Extra, untimed pass with the streaming Adam back on the main stream: clean per-kernel durations (in the timed
    # region it overlaps the layer kernels on a side stream, which stretches both sides' event intervals).
    phases_serial = {}
    if not args.no_phase_timing and eng.overlap and not args.train_only:
        eng.overlap = False
        eng.timers = {}
        for i in range(W, W + min(K, 5)):
            step(i)
        eng.flush_lazy()
        phases_serial = eng.phase_ms()
        eng.timers = None
        eng.overlap = True

    # ---- sustained leg (single rank): thousands of steps over hundreds of DISTINCT batches, flush inside - long enough for a
    #      5-second utilisation sampler to see, and the steady state of a real epoch (the 20-step headline pays one flush per
    #      20 steps; an epoch pays one per thousands) -------------------------------------------------------------------------
    sustained = None
    n_sus = args.sustained_steps if args.sustained_steps is not None else (2000 if args.config != "c5" else 120)
    if world == 1 and n_sus > 0 and not args.train_only:
        n_distinct = max(1, min(200 if args.config != "c5" else 30, n_sus))
        Xs, ys = synth_batches(n_distinct * B, seed=4242, ids=args.ids)
        Xsd, ysd = torch.from_numpy(Xs).to(device), torch.from_numpy(ys).to(device)
        eng.timers = None
        for i in range(3):
            eng.train_step(Xsd[i * B:(i + 1) * B], ysd[i * B:(i + 1) * B])
        eng.flush_lazy()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sus_timers = {}
        for i in range(n_sus):
            j, jn = i % n_distinct, (i + 1) % n_distinct
            eng.timers = sus_timers if (i % 50 == 49 and not args.no_phase_timing) else None   # a phase sample every 50th step
            eng.train_step(Xsd[j * B:(j + 1) * B], ysd[j * B:(j + 1) * B], next_X=Xsd[jn * B:(jn + 1) * B])
        eng.timers = sus_timers if not args.no_phase_timing else None
        eng.flush_lazy()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        sus_ph = eng.phase_ms() if eng.timers is not None else {}
        n_fl = len((eng.timers or {}).get("lazy_flush", []))
        eng.timers = None
        sustained = {"steps": n_sus, "distinct_batches": n_distinct, "ms_per_step": round(dt / n_sus * 1e3, 4),
                     "samples_per_s": round(n_sus * B / dt, 1), "wall_s": round(dt, 2),
                     "phase_ms_per_launch": {k: round(v, 4) for k, v in sus_ph.items()},
                     "lazy_flush": {"ms_per_launch": round(sus_ph.get("lazy_flush", 0.0), 4), "every_steps": eng.flush_every,
                                    "ms_per_step": round(sus_ph.get("lazy_flush", 0.0) / max(1, eng.flush_every), 4),
                                    "launches_timed": n_fl,
                                    "note": "a flush replays flush_every postponed steps of every row not gathered since (the "
                                            f"headline's one flush replays its K = {K} steps, this one {eng.flush_every}: the same work "
                                            "per step)"},
                     "note": "same step as `value`, lazy flush inside the timed region, one rank; phases sampled every 50th step "
                             "(a row's postponed optimizer steps are replayed when it is next gathered: the replay chains are "
                             "longer here than in the 20-step headline)"}
        del Xsd, ysd

    # ---- fit()-level leg: the public API as reference main.py drives it (dict of columns, verbose=1 with its per-step
    #      binary_crossentropy + auc, shuffle=True), everything fit does per epoch included (upload, shuffle index_select,
    #      device metrics, History) ------------------------------------------------------------------------------------------------
    fit_leg = None
    n_fit = args.fit_batches if args.fit_batches is not None else (200 if args.config != "c5" else 0)
    if world == 1 and n_fit > 0 and not args.train_only:
        try:
            Xf, yf = synth_batches(n_fit * B, seed=777, ids=args.ids)
            names = list(CFG["fields"]) + list(CFG["dense"])
            xf = {f: (Xf[:, i].astype(np.int64) if i < len(CFG["fields"]) else Xf[:, i]) for i, f in enumerate(names)}
            # (a four-batch fit first: the first call of the per-step metric ops pays their one-time initialisation, ~1 s)
            model.fit(x={k: v[:4 * B] for k, v in xf.items()}, y=yf[:4 * B], batch_size=B, epochs=1, verbose=1, shuffle=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hist = model.fit(x=xf, y=yf, batch_size=B, epochs=1, verbose=1, shuffle=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            # the same call with two epochs: the second one finds the dataset resident (no packing, no upload)
            t2 = time.perf_counter()
            model.fit(x=xf, y=yf, batch_size=B, epochs=2, verbose=1, shuffle=True)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
            warm = max(dt2 - dt, 1e-9)
            fit_leg = {"rows": int(n_fit * B), "batch_size": B, "wall_s": round(dt, 3), "samples_per_s": round(n_fit * B / dt, 1),
                       "warm_epoch": {"wall_s": round(warm, 3), "samples_per_s": round(n_fit * B / warm, 1),
                                      "note": "a two-epoch fit minus the one-epoch fit: an epoch over the already resident dataset"},
                       "epoch_loss": float(hist.history["loss"][-1]) if hist.history.get("loss") else None,
                       "note": "model.fit(x=dict, y, batch_size, epochs=1, verbose=1, shuffle=True): upload of the dataset's columns, the "
                               "epoch's shuffle order (the reference loader's permutation), per-step train metrics "
                               "(binary_crossentropy, auc), History"}
            model.train()
        except Exception as ex:
            print(f"[bench] fit leg skipped: {ex}", file=sys.stderr)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel ----------------------------------------------------------------
    F, D, L, U = len(CFG["fields"]), CFG["D"], CFG["L"], CFG["units"][0]
    total_rows = model.embedding_arena.shape[0]
    uniq = int(torch.unique(eng._ws[B]["rows"]).numel()) if B in eng._ws else 0
    # per layer launch: projections + output 4 F D^2, MetaNet of Q and K 4 F D U (U = 2 D: 12 F D^2 in all),
    # attention 2 F^2 D MACs per sample; flag `gate` has no MetaNet (an elementwise product), `bilinear` a d x d map per head of q
    meta_macs = 0 if "gate" in args.flag else (F * D * (D // CFG["H"]) if "bilinear" in args.flag else 4 * F * D * U)
    fwd_flops = 2.0 * (4 * F * D * D + meta_macs + 2 * F * F * D) * B
    # which kernels a "layer_fwd" / "layer_bwd" phase of THIS configuration consists of: one fused kernel (+ its reduction
    # launch) on the fused path; on the general path (configs[4], gate / bilinear) a phase is a chain of ~25 / ~45 launches
    # of the gen_* families, so the phase's rate is a whole-layer rate and no single kernel's counters describe it
    generic = bool(eng._ws.get(B, {}).get("generic"))
    one_kernel = not generic
    fwd_kernel = "layer_fwd_fused_kernel" if one_kernel else "general path, whole layer forward (~25 gen_* launches)"
    bwd_kernel = eng.bwd_kernel_name() if one_kernel else "general path, whole layer backward (~45 gen_* launches)"
    # the last layer of a step as ONE launch with the head fused in (satrans_layer_bwd_head): its recomputed forward IS the
    # layer's forward (no forward launch exists for it), so the launch carries forward + backward = 3 x the forward's FLOPs
    fused_head = bool(eng._ws.get(B, {}).get("fuse_head"))
    saved_attn = any(t is not None for t in (eng._ws.get(B, {}).get("attn_save") or []))
    per_launch = {
        "layer_bwd_head": dict(kernel=bwd_kernel + " (HEADF instantiation: last layer forward + head + loss + backward)",
                               bound="mfma", unit="TFLOP/s", peak=FP32_PEAK_TFLOPS, work=3.0 * fwd_flops / 1e12),
        "adam_untouched": dict(kernel="adam_untouched_kernel", bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS,
                               work=6.0 * (total_rows - uniq * world) * D * 4 / 1e9),   # read p,m,v + write p,m,v
        "layer_bwd": dict(kernel=bwd_kernel, bound="mfma", unit="TFLOP/s", peak=FP32_PEAK_TFLOPS, work=2.0 * fwd_flops / 1e12),
        "layer_fwd": dict(kernel=fwd_kernel, bound="mfma", unit="TFLOP/s", peak=FP32_PEAK_TFLOPS, work=fwd_flops / 1e12),
        "layer_fwd_gather": dict(kernel=fwd_kernel + " (layer 0: tokens read from the embedding arena, gather fused in)", bound="mfma",
                                 unit="TFLOP/s", peak=FP32_PEAK_TFLOPS, work=fwd_flops / 1e12),
        "gather_fwd": dict(kernel="gather_rows_kernel", bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS,
                           work=B * F * (D * 4 + 4) / 1e9),                        # read bytes (only timed with engine.fuse_gather off)
    }
    n_flush = max(1, n_flush_timed)
    n_sep = L - 1 if fused_head else L
    gather_fused = "layer_fwd_gather" in phases or "layer_fwd_gather" in dispatch
    count = {"layer_fwd": n_sep - (1 if gather_fused else 0), "layer_fwd_gather": 1, "layer_bwd": n_sep, "layer_bwd_head": 1,
             "lazy_flush": n_flush / K}  # launches per step (the flush runs every
    #                                   SATRANS_LAZY_FLUSH_EVERY = 32 steps and once more at the end of the timed region)

    def table(ph, count=count, disp=None):
        # `disp`: phase -> the kernel's own duration (events signalled by the dispatch, read_dispatch_ms): the fused layer kernels,
        # which are then NOT bracketed by recorded events (two markers less per launch in the queue: `timed_by` says which)
        rows, dom, dom_time = {}, None, -1.0
        order = [n for n in LAYER_PHASES if disp and n in disp and n not in ph] + list(ph)
        for name in order:
            ms = ph[name] if name in ph else disp[name]
            per_step = ms * count.get(name, 1)
            entry = {"ms_per_launch": round(ms, 4), "ms_per_step": round(per_step, 4)}
            if name not in ph:
                entry["timed_by"] = "dispatch"
            elif disp and name in disp:
                entry["kernel_ms"] = round(disp[name], 4)
            if name == "lazy_flush":
                entry["note"] = (f"all {total_rows:,} rows brought up to date {n_flush} time(s) inside the timed region of {K} steps "
                                 f"(every {eng.flush_every} steps and at its end)")
            if name in per_launch:
                spec = per_launch[name]
                ach = spec["work"] / (entry.get("kernel_ms", ms) / 1e3)
                entry.update(bound=spec["bound"], achieved=round(ach, 2), peak=spec["peak"], unit=spec["unit"],
                             frac=round(ach / spec["peak"], 4))
                if per_step > dom_time:
                    dom, dom_time = name, per_step
            rows[name] = entry
        return rows, dom

    kernels, dominant = table(phases, disp=dispatch)
    phase_sum = sum(v["ms_per_step"] for v in kernels.values())
    kernels_serial, _ = table(phases_serial)
    roofline = None
    if dominant:                              # the phase with the largest time per step inside the timed region
        e, spec = kernels[dominant], per_launch[dominant]
        roofline = {"kernel": spec["kernel"], "bound": e["bound"], "achieved": e["achieved"], "peak": e["peak"],
                    "unit": e["unit"], "frac": e["frac"], "traffic": None,
                    "algorithmic_per_launch": spec["work"], "launch_ms": e.get("kernel_ms", e["ms_per_launch"]),
                    "timed_by": e.get("timed_by", "recorded events"),
                    "launches_per_step": count.get(dominant, 1),
                    # HBM bytes the launch has to move (what `traffic` is to be read against).  `_min`: layer input, upstream
                    # gradient and input gradient [B,F,D] each - what a backward that recomputes everything moves.  `_per_launch`
                    # adds the state the forward handed over (softmax numerators H F F, 1 / sum and keep word H F, attention
                    # output F D, the normalised MetaNet rows of both roles 2 F D and their 1 / std 2 F per sample), which the
                    # backward reads INSTEAD of recomputing it: bytes traded for matrix-pipe and VALU time in a compute-bound kernel
                    "algorithmic_bytes_min": 3 * B * F * D * 4 if dominant in ("layer_bwd", "layer_bwd_head") else None,
                    "algorithmic_bytes_per_launch": (3 * B * F * D * 4 + (B * (CFG["H"] * F * F + 2 * CFG["H"] * F + F * D + F * (2 * D + 2)) * 4
                                                                         if (dominant == "layer_bwd" and saved_attn) else 0))
                                                    if dominant in ("layer_bwd", "layer_bwd_head") else None,
                    "note": "timed_by dispatch: `launch_ms` is the kernel's own duration - HIP events that its dispatch signals with its "
                            "begin / end timestamps (hipExtLaunchKernelGGL through satrans_kernel_timing), on the stream it runs on, "
                            "every 4th step of the timed region (two events RECORDED around a launch add the dispatch latency on "
                            "either side, 5-15 us here, and two markers to the queue; of the L - 1 plain backward launches of a step the last one - "
                            "layer 0, the same kernel - goes untimed, because the next batch's preparation forks in front of it and "
                            "must not find a marker there); the other phases are timed by recorded events; " +
                            ("the weight-gradient slabs of all layers are reduced by ONE launch per step (`layer_bwd_reduce`), "
                             "so a layer_bwd launch is the backward kernel alone; the last layer runs as `layer_bwd_head` "
                             "(its forward, the head, the loss and their backward in one launch)" if "layer_bwd_reduce" in phases
                             else "layer_bwd includes its fixed-order reduction launch") +
                            "; `kernels_serial` repeats the measurement in an extra untimed pass"}

    # HBM bytes per launch of that kernel from the PMC passes under profiles/ (tools/pmc_passes.sh: FETCH_SIZE and WRITE_SIZE
    # in separate passes of `bench.py --train-only`, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  The
    # summary records the hash of the kernel SOURCES it profiled and the bench configuration: a summary of other sources or
    # of another configuration is reported as such, never as this run's traffic; a phase made of many kernels has none.
    if roofline and not one_kernel:
        roofline["traffic_source"] = "none: the phase is a chain of launches (general path); see profiles/ for per-kernel stats"
        try:
            from satrans_amd import native as _native
            name = PMC_SUMMARIES.get(args.config)
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            key = {"layer_bwd": "layer_bwd_chain", "layer_fwd": "layer_fwd_chain", "layer_fwd_gather": "layer_fwd_chain"}[dominant]
            if pmc.get("_source_sha256") != _native.source_hash():
                roofline["traffic_source"] = (f"stale: profiles/{name} was taken on other kernel sources "
                                              f"({str(pmc.get('_source_sha256'))[:12]} vs {_native.source_hash()[:12]})")
            elif args.flag == CFG["flag"] and not CFG.get("scaled"):
                roofline["traffic"] = int(pmc[key]["bytes_per_launch"])
                roofline["traffic_ratio_min"] = round(roofline["traffic"] / roofline["algorithmic_bytes_min"], 3)
                roofline["traffic_source"] = (f"profiles/{name}: rocprofv3 --pmc of these kernel sources, 2 x FETCH_SIZE + WRITE_SIZE summed "
                                              f"over the launches of one layer's chain (activations live in HBM between the launches of "
                                              f"the general path, so the ratio to x + dy + dx is the price of not fusing)")
        except (OSError, KeyError, TypeError):
            pass
    elif roofline:
        try:
            from satrans_amd import native as _native
            sha = _native.source_hash()
            pmc_key = args.config if args.flag == CFG["flag"] else f"{args.config}:{args.flag}"
            PMC_SUMMARY = PMC_SUMMARIES.get(pmc_key, PMC_SUMMARIES["aliccp"])
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_SUMMARY)))
            rec = pmc[{"layer_bwd": "layer_bwd_fused_kernel", "layer_bwd_head": "layer_bwd_fused_kernel[head]",
                       "layer_fwd": "layer_fwd_fused_kernel", "layer_fwd_gather": "layer_fwd_fused_kernel @layer0"}.get(dominant, roofline["kernel"])]
            if pmc.get("_source_sha256") != sha:
                roofline["traffic_source"] = (f"stale: profiles/{PMC_SUMMARY} was taken on other kernel sources "
                                              f"({str(pmc.get('_source_sha256'))[:12]} vs {sha[:12]})")
            elif pmc.get("_config") != pmc_key:
                roofline["traffic_source"] = f"none: profiles/{PMC_SUMMARY} is of config {pmc.get('_config')!r}, default flag"
            else:
                roofline["traffic"] = round((2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0)
                if roofline.get("algorithmic_bytes_min"):
                    roofline["traffic_ratio_min"] = round(roofline["traffic"] / roofline["algorithmic_bytes_min"], 3)
                    roofline["traffic_ratio"] = round(roofline["traffic"] / roofline["algorithmic_bytes_per_launch"], 3)
                roofline["traffic_source"] = (f"profiles/{PMC_SUMMARY} (rocprofv3 --pmc of these kernel sources, bytes per "
                                              f"launch: 2 x FETCH_SIZE + WRITE_SIZE)")
        except (OSError, StopIteration, KeyError):
            pass

    # ---- the gather on its own: achieved HBM GB/s at the training batch and at the reference's prediction batch
    #      (main.py:353 predicts with 4 x batch_size), a different id batch for every launch -----------------------------
    gather = gather_microbench(eng, Xd, B, F, D) if (not args.train_only or args.with_gather) else None
    fwd_ms = {**dispatch, **phases}
    if gather is not None and gather_fused and "layer_fwd" in fwd_ms:
        # In the training step no gather kernel runs: layer 0 reads its B x F rows (128-byte random reads) straight from the arena.
        # What that costs is the difference between layer 0's forward launch and the same kernel on dense [B,F,D] input (layer 1).
        exposed = fwd_ms["layer_fwd_gather"] - fwd_ms["layer_fwd"]
        nbytes = B * F * (D * 4 + 4)
        # (the training step's answer first: what the step pays for the gather)
        gather = {"exposed_us_in_training_step": round(1e3 * exposed, 2), **gather}
        gather["fused_into_layer0"] = {
            "layer0_fwd_ms": round(fwd_ms["layer_fwd_gather"], 4), "other_layer_fwd_ms": round(fwd_ms["layer_fwd"], 4),
            "exposed_ms": round(exposed, 4), "algorithmic_read_bytes": nbytes,
            "hidden": bool(exposed <= 0.002),
            "read_GBps_if_all_exposed_time_were_the_gather": round(nbytes / 1e9 / (max(exposed, 1e-4) / 1e3), 1),
            "note": "timed region of the training step, HIP events on every 4th step: the gather's random row reads run underneath "
                    "layer 0's arithmetic (the kernel is compute-bound); `exposed_ms` is all of it that the step pays for"}

    # ---- the evaluation forward alone (predict / evaluate path) at the reference's prediction batch --------------------------
    forward_only = None
    try:
        if args.train_only:
            raise RuntimeError("--train-only")
        model.eval()
        nb = min(4 * B, Xd.shape[0])
        for _ in range(3):
            eng.forward(Xd[:nb])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for i in range(reps):
            lo = (i * nb) % max(1, Xd.shape[0] - nb + 1)
            eng.forward(Xd[lo:lo + nb])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = L * fwd_flops * (nb / B) / 1e12                                      # all layers of the evaluation forward, TFLOP
        forward_only = {"batch": nb, "ms_per_batch": round(ms, 4), "samples_per_s": round(nb / (ms / 1e3), 1),
                        "roofline": {"bound": "mfma", "unit": "TFLOP/s", "achieved": round(fl / (ms / 1e3), 2), "peak": FP32_PEAK_TFLOPS,
                                     "frac": round(fl / (ms / 1e3) / FP32_PEAK_TFLOPS, 4),
                                     "note": "whole evaluation forward (gather + L layer launches + head) between two events"}}
        model.train()
    except Exception as ex:
        print(f"[bench] forward-only timing skipped: {ex}", file=sys.stderr)

    # ---- the same evaluation forward with the dense products on the bf16 matrix pipe (BASELINE configs[1] "bf16 forward"):
    #      its own line, never the headline; logit error against the fp32 kernels on the same inputs ---------------------------
    forward_bf16 = None
    try:
        if args.train_only:
            raise RuntimeError("--train-only")
        model.eval()
        nb = min(4 * B, Xd.shape[0])
        model(Xd[:2048])
        l32 = eng.last_logit().clone()
        model.set_forward_precision("bf16")
        model(Xd[:2048])
        lb16 = eng.last_logit().clone()
        for _ in range(3):
            eng.forward(Xd[:nb])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for i in range(reps):
            lo = (i * nb) % max(1, Xd.shape[0] - nb + 1)
            eng.forward(Xd[lo:lo + nb])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = L * fwd_flops * (nb / B) / 1e12
        # which pipe bounds it: 12 F D^2 of the 12 F D^2 + 2 F^2 D MACs per sample and layer run as bf16 MFMA (16 x the fp32 rate),
        # the attention dots, softmax and LayerNorms stay fp32 VALU - the fp32 vector pipe is the bound, so the fraction is quoted
        # against BOTH peaks: the kernel's fp32 work against the fp32 peak, all of its work against the bf16 dense peak
        fl32 = L * 2.0 * (2 * F * F * D) * nb / 1e12
        forward_bf16 = {"dtype": "bf16", "batch": nb, "ms_per_batch": round(ms, 4), "samples_per_s": round(nb / (ms / 1e3), 1),
                        "roofline": {"bound": "valu-f32 (attention, softmax, LayerNorm); products on bf16 mfma", "unit": "TFLOP/s",
                                     "achieved": round(fl / (ms / 1e3), 2), "peak": BF16_PEAK_TFLOPS,
                                     "frac": round(fl / (ms / 1e3) / BF16_PEAK_TFLOPS, 4),
                                     "fp32_part_achieved": round(fl32 / (ms / 1e3), 2), "fp32_peak": FP32_PEAK_TFLOPS,
                                     "fp32_part_frac": round(fl32 / (ms / 1e3) / FP32_PEAK_TFLOPS, 4),
                                     "equivalent_frac_of_fp32_peak": round(fl / (ms / 1e3) / FP32_PEAK_TFLOPS, 4)},
                        "logit_max_abs_err_vs_fp32_kernels": float((lb16 - l32).abs().max()),
                        "note": "evaluation forward only; products bf16 x bf16 -> fp32 (v_mfma_f32_16x16x32_bf16), LayerNorm / "
                                "softmax / attention dots fp32; training stays fp32"}
        model.set_forward_precision("fp32")
        model.train()
    except Exception as ex:
        print(f"[bench] bf16 forward timing skipped: {ex}", file=sys.stderr)
        try:
            model.set_forward_precision("fp32")
        except Exception:
            pass

    # ---- parity figure the metric asks for: forward logits vs the CPU oracle on identical inputs -------------
    err = err_train = logit_scale = None
    try:
        if args.train_only:
            raise RuntimeError("--train-only")
        from oracle import satrans_oracle as O
        model.eval()
        nb = 2048
        model(Xd[:nb])
        gpu_logit = eng.last_logit().cpu()
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        _, ref_logit = O.forward(sd, torch.from_numpy(X[:nb]), oracle_spec(args.flag))
        err = float((gpu_logit - ref_logit).abs().max())
        # the same inputs through the TRAINING forward (the kernels of the training step) with its dropouts switched off
        model.train()
        keep_p, eng.drop_p = eng.drop_p, 0.0
        try:
            model(Xd[:nb])
            err_train = float((eng.last_logit().cpu() - ref_logit).abs().max())
        finally:
            eng.drop_p = keep_p
            model.eval()
        logit_scale = float(ref_logit.abs().max())
        del sd
    except Exception as ex:                                                        # the number is informative only
        print(f"[bench] logit parity check skipped: {ex}", file=sys.stderr)

    cpu = None
    if do_cpu:
        t_cpu = time.time()
        n_need = 1 + 3 * 5 + 2 + args.cpu_steps + 5 + 2      # first step, thread sweep (3 per candidate), warm-up, timed, verbose
        # bounded sample: the same step at a smaller batch where one CPU step of the full batch would take minutes (c5)
        Bc = B if args.config != "c5" else min(B, 512)
        timed_c = args.cpu_steps if args.config != "c5" else min(args.cpu_steps, 3)
        Xc, yc = synth_batches(n_need * Bc, seed=7, cfg=cpu_cfg)
        r = cpu_baseline(state_cpu, Xc, yc, Bc, args.lr, args.flag, timed=timed_c, timed_verbose=5 if args.config != "c5" else 2)
        cpu = {"value": round(r["value"], 1), "unit": "samples/s", "cores": r["threads"], "threads": r["threads"], "kind": "port",
               "cpu_model": cpu_model_name(), "host_cores": os.cpu_count(),
               "value_verbose1": round(r["value_verbose1"], 1),
               "thread_sweep_s_per_step": r["sweep_s_per_step"],
               "sample": f"median of {r['timed']} training steps of B={Bc} after 2 untimed ones at the best thread count of a "
                         f"sweep (median of 3 steps per candidate; dropout on, dense L2 + dense torch Adam over all {n_cpu_rows:,} rows; `value` = verbose=0, "
                         f"`value_verbose1` = median of {r['timed_verbose']} steps with the per-step sklearn log_loss + "
                         f"roc_auc_score of fit(verbose=1), what reference main.py runs); {time.time() - t_cpu:.0f}s wall"}

    # ---- the other BASELINE configs and variants on the same clock: one child process each (a fresh engine, nothing shared with
    #      this one), 20 timed steps after 5, train-only; this process only relays a compact record of each line ---------------
    others = None
    if world == 1 and not args.train_only and not args.no_other_configs and args.config == "aliccp" and args.flag == "sota" \
            and args.ids == "uniform":
        others = other_configs()

    value = world * B * K / elapsed
    out = {
        "metric": ("training samples/sec (AliCCP-shaped, emb=32, 3L/4H, meta_mode=QK)" if args.config == "aliccp" else
                   f"training samples/sec ({args.config}-shaped{' SCALED tables' if CFG.get('scaled') else ''}, emb={D}, {L}L/{CFG['H']}H, meta_mode=QK)") +
                  ("" if args.flag == "sota" else f" flag={args.flag}"),
        "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "products": "f32: v_mfma_f32_16x16x4_f32 (bit for bit an fmaf chain)",
        "config": {"workload": f"{CFG['label']}, {model.embedding_arena.shape[0]:,} table rows "
                               f"({model.embedding_arena.numel() * 4 / 1e6:,.0f} MB fp32), {args.ids} ids, dropout on, "
                               f"dense-Adam+L2 semantics over all rows",
                   "batch_per_gpu": B, "global_batch": B * world, "embedding_dim": D, "layers": L, "heads": CFG["H"],
                   "fields": F, "parallelism": f"dp{world}"},
        "fwd_logit_max_abs_err_vs_cpu_oracle": err,
        "fwd_logit": {"max_abs_err_vs_cpu_oracle": err, "training_forward_dropout_off_max_abs_err": err_train,
                      "max_abs_logit": logit_scale, "samples": 2048,
                      "note": "weights as they are after this run's training steps; evaluation and training forward"},
        "sustained_ms_per_step": sustained["ms_per_step"] if sustained else None, "sustained": sustained,
        "fit_samples_per_s": fit_leg["samples_per_s"] if fit_leg else None, "fit": fit_leg,
        "phase_sum_ms_per_step": round(phase_sum, 4), "phase_sum_frac_of_step": round(phase_sum / (elapsed / K * 1e3), 4),
        "roofline": roofline, "kernels": kernels, "kernels_serial": kernels_serial, "gather": gather,
        "forward_only": forward_only, "forward_only_bf16": forward_bf16, "cpu_baseline": cpu, "other_configs": others,
        "per_rank_ms_per_step": [round(v, 4) for v in per_rank_ms], "collectives": collectives,
    }
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
