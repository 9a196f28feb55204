"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py          # writes tests/golden/

The reference (`/root/reference`, read-only, never copied) is imported with the stand-in packages
of `oracle/shims/` on sys.path.  For every case below this script builds the reference's
`models.satrans.SATrans`, feeds it seeded synthetic inputs and records, as plain arrays:

  param/<key>      every unique tensor of state_dict() right after construction (seed parity)
  alias/<key>      for keys that share storage with another key: the name of that key
  X, y             the float32 input matrix in `feature_index` column order, and labels
  out/prob|logit   eval-mode forward output and the pre-sigmoid logit
  out/att_input, out/vec0, out/q{l}, out/k{l}, out/att{l}, out/layer{l}
  train/bce|reg    loss pieces of ONE train-mode step with every dropout probability set to 0
  grad/<key>       gradients of that step (keys without gradient are absent)
  adam/<key>       every unique tensor after `adam_steps` steps of torch.optim.Adam at `lr`
  opt/exp_avg/<key>, opt/exp_avg_sq/<key>   torch.optim.Adam's moments after those steps
  fit/*            History['loss'] and predict() output of the reference's own fit()/predict()

Nothing here runs on the GPU box; the fixtures are data (inputs + expected outputs).
"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(ROOT, "oracle", "shims"), ROOT, "/root/reference"]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from deepctr_torch.inputs import SparseFeat, DenseFeat, get_feature_names  # noqa: E402  (shim)
from models.satrans import SATrans  # noqa: E402  (the reference)

ALICCP_FIELDS = ['101', '121', '122', '124', '125', '126', '127', '128', '129', '205', '206', '207', '210',
                 '216', '508', '509', '702', '853', '301']              # reference main.py:99-101
ALICCP_MAX = {'101': 444861, '121': 97, '122': 13, '124': 2, '125': 7, '126': 3, '127': 3, '128': 2, '129': 4,
              '205': 4348615, '206': 8993, '207': 695124, '210': 99606, '216': 234880, '508': 8185,
              '509': 472354, '702': 167813, '853': 91358, '301': 3}    # reference main.py:124-127
ALIMAMA_FIELDS = ['user_id', 'adgroup_id', 'pid', 'cms_segid', 'cms_group_id', 'final_gender_code',
                  'age_level', 'pvalue_level', 'shopping_level', 'occupation', 'new_user_class_level',
                  'cate_id', 'campaign_id', 'customer', 'brand']       # reference main.py:143-145
ALIMAMA_MAX = {'user_id': 60, 'adgroup_id': 50, 'pid': 1, 'cms_segid': 40, 'cms_group_id': 12,
               'final_gender_code': 2, 'age_level': 6, 'pvalue_level': 3, 'shopping_level': 3, 'occupation': 1,
               'new_user_class_level': 4, 'cate_id': 45, 'campaign_id': 55, 'customer': 35, 'brand': 48}


def make_case(name):
    """-> dict(fields, maxima, dense, domain, D, H, L, units, flag, mode, lr, B, seed, cap)"""
    base = dict(fields=ALICCP_FIELDS, maxima=ALICCP_MAX, dense=[], domain=['301'], D=32, H=4, L=3,
                units=(64, 32), flag='sota', mode='QK', lr=0.005, B=64, seed='1021', cap=40,
                adam_steps=3, with_fit=False, train=True)
    small = dict(base, fields=['f0', 'f1', 'f2', 'f3', 'f4', 'dom'],
                 maxima={'f0': 30, 'f1': 9, 'f2': 2, 'f3': 17, 'f4': 5, 'dom': 3}, domain=['dom'],
                 D=16, H=2, L=2, units=(32, 16), B=48, adam_steps=2)
    cases = {
        'aliccp_sota': dict(base, with_fit=True),
        'alimama_sota_pos': dict(base, fields=ALIMAMA_FIELDS, maxima=ALIMAMA_MAX, dense=['price'],
                                 domain=['shopping_level'], flag='sota-pos', lr=0.001, with_fit=True),
        'small_q': dict(small, mode='Q'),
        'small_query': dict(small, mode='Query'),          # main.py:59 default; substring test => Q only
        'small_k': dict(small, mode='K'),
        'small_qkv': dict(small, mode='QKV'),
        'small_none': dict(small, mode='V'),               # no modulation at all
        'small_gate': dict(small, flag='sota-gate'),
        'small_bilinear': dict(small, flag='sota-bilinear'),
        'small_relu': dict(small, flag='sota-relu', train=False),   # see run_case: p=0 dropout + in-place residual
                                                                  # cannot backprop through the reference's ReLU branch
        'small_onlyemb': dict(small, flag='sota-onlyemb'),
        'small_pos_dense': dict(small, flag='sota-pos', dense=['price', 'age']),
        'small_d64': dict(small, D=64, H=4, L=1, units=(16, 64), B=24),
        'small_multidomain': dict(small, domain=['dom', 'f2']),
        # the shape class of BASELINE configs[4] (embedding_dim 64, MetaNet hidden 128, more fields than one 32-key chunk):
        # 40 fields, units (128, 64) - served by the general layer path (csrc/layer_generic.hip)
        'small_d64_u128': dict(small, fields=[f'g{i}' for i in range(39)] + ['dom'],
                               maxima=dict({f'g{i}': 2 + (5 * i) % 11 for i in range(39)}, dom=3),
                               D=64, H=4, L=2, units=(128, 64), B=24, save_adam=False),   # (its encoder weight alone is 4 MB)
    }
    return cases[name], list(cases)


def synth_inputs(cfg, rng, B):
    cols = {}
    for f in cfg['fields']:
        hi = min(cfg['maxima'][f], cfg['cap'])
        lo = 1 if f == cfg['domain'][0] and f in ('301',) else 0     # AliCCP scenario ids start at 1 (main.py:112-114)
        cols[f] = rng.randint(lo, hi + 1, size=B).astype(np.int64)
    for f in cfg['dense']:
        cols[f] = rng.rand(B).astype(np.float32)
    y = (rng.rand(B) < 0.3).astype(np.float32)
    return cols, y


def build_reference(cfg):
    vocab = {f: min(cfg['maxima'][f], cfg['cap']) + 2 for f in cfg['fields']}    # main.py:182 `data_max + 2`
    columns = [SparseFeat(f, vocabulary_size=vocab[f], embedding_dim=cfg['D']) for f in cfg['fields']] + \
              [DenseFeat(f, 1) for f in cfg['dense']]
    num_domains_list = [min(cfg['maxima'][c], cfg['cap']) for c in cfg['domain']]
    model = SATrans(linear_feature_columns=columns, dnn_feature_columns=columns,
                    domain_column_list=list(cfg['domain']), num_domains_list=num_domains_list,
                    att_layer_num=0, domain_att_layer_num=cfg['L'], att_head_num=cfg['H'],
                    share_domain_dnn_across_layers=False, use_domain_dnn_linear=False, use_linear=False,
                    meta_mode=cfg['mode'], use_dnn=False, meta_dnn_hidden_units=cfg['units'],
                    seed=cfg['seed'], device='cpu', flag=cfg['flag'])             # main.py:292-306
    return model, columns, vocab, num_domains_list


def pack_state(model, prefix, out):
    seen = {}
    for k, t in model.state_dict().items():
        ptr = t.data_ptr()
        if ptr in seen:
            out[f"alias/{k}"] = np.array(seen[ptr])
        else:
            seen[ptr] = k
            out[f"{prefix}/{k}"] = t.detach().cpu().numpy().copy()


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0


def run_case(name, outdir):
    cfg, _ = make_case(name)
    rng = np.random.RandomState(sum(map(ord, name)))
    model, columns, vocab, num_domains_list = build_reference(cfg)
    out = {}
    pack_state(model, "param", out)
    init_state = {k: v.clone() for k, v in model.state_dict().items()}

    B = cfg['B']
    cols, y = synth_inputs(cfg, rng, B)
    names = get_feature_names(columns)
    X = np.concatenate([np.asarray(cols[n]).reshape(B, -1) for n in names], axis=-1).astype(np.float32)
    Xt = torch.from_numpy(X)
    out["X"], out["y"] = X, y

    # ---- eval forward with intermediates captured through hooks --------------------------------
    taps = {}
    def tap_logit(m, i, o):
        taps["logit"] = o.detach().clone()

    def make_layer_tap(l):
        def tap(m, i, o):
            taps[f"layer{l}"] = o.detach().clone()
            taps[f"att{l}"] = m.normalized_att_scores.detach().clone()
            taps[f"vec{l}"] = i[1].detach().clone()
        return tap

    def tap_input(m, i):
        taps["att_input"] = i[0].detach().clone()

    hooks = [model.dnn_linear.register_forward_hook(tap_logit)]
    for l, layer in enumerate(model.domain_int_layers):
        hooks.append(layer.register_forward_hook(make_layer_tap(l)))
        if l == 0:
            hooks.append(layer.register_forward_pre_hook(tap_input))
        calls = {"n": 0}

        def meta_tap(m, i, o, l=l, layer=layer, calls=calls):
            # Q_meta_mlp is called for Q first, then (when aliased) for K (models/satrans.py:60-73)
            if m is layer.Q_meta_mlp and m is layer.K_meta_mlp:
                order = []
                if 'Q' in cfg['mode']:
                    order.append('q')
                if 'K' in cfg['mode']:
                    order.append('k')
                taps[f"{order[calls['n'] % len(order)]}{l}"] = o.detach().clone()
                calls['n'] += 1
            elif m is layer.Q_meta_mlp:
                taps[f"q{l}"] = o.detach().clone()
            else:
                taps[f"k{l}"] = o.detach().clone()
        hooks.append(layer.Q_meta_mlp.register_forward_hook(meta_tap))
        if layer.K_meta_mlp is not layer.Q_meta_mlp:
            hooks.append(layer.K_meta_mlp.register_forward_hook(meta_tap))
    model.eval()
    with torch.no_grad():
        prob = model(Xt)
    for h in hooks:
        h.remove()
    out["out/prob"] = prob.numpy().copy()
    for k, v in taps.items():
        if k.startswith("vec") and k != "vec0":
            continue
        out[f"out/{k}"] = v.numpy()[:8].copy() if k == "vec0" else v.numpy().copy()

    # ---- one train-mode step, dropout p = 0: loss pieces and gradients -----------------------------
    zero_dropout(model)
    model.train()
    optim = torch.optim.Adam(model.parameters(), lr=cfg['lr'])                    # main.py:343
    model.compile(optim, "binary_crossentropy", metrics=["binary_crossentropy", "auc"])
    yt = torch.from_numpy(y)
    for step in range(cfg['adam_steps'] if cfg['train'] else 0):
        y_pred = model(Xt).squeeze()
        optim.zero_grad()
        loss = model.loss_func(y_pred, yt, reduction='sum')                       # meta_basemodel.py:317
        reg = model.get_regularization_loss()                                     # :318
        total = loss + reg + model.aux_loss
        total.backward()
        if step == 0:
            out["train/bce"] = np.array(loss.item(), dtype=np.float64)
            out["train/reg"] = np.array(reg.item(), dtype=np.float64)
            seen = set()
            for k, p in model.named_parameters():
                if p.grad is not None and p.data_ptr() not in seen:
                    seen.add(p.data_ptr())
                    out[f"grad/{k}"] = p.grad.detach().numpy().copy()
        optim.step()
    if cfg['train'] and cfg.get('save_adam', True):
        pack_state(model, "adam", out)
        # torch.optim.Adam's own state after those steps: well-conditioned where the parameters are not (a parameter
        # element with |g| ~ eps moves by up to lr per step whatever its moments' last bits are), so these pin the
        # optimizer trajectory tightly
        seen = set()
        for k, p in model.named_parameters():
            if p in optim.state and p.data_ptr() not in seen and 'exp_avg' in optim.state[p]:
                seen.add(p.data_ptr())
                out[f"opt/exp_avg/{k}"] = optim.state[p]['exp_avg'].detach().numpy().copy()
                out[f"opt/exp_avg_sq/{k}"] = optim.state[p]['exp_avg_sq'].detach().numpy().copy()

    # ---- the reference's own fit()/predict() on a fresh model -------------------------------------
    if cfg['with_fit']:
        model2, _, _, _ = build_reference(cfg)
        assert all(torch.equal(init_state[k], v) for k, v in model2.state_dict().items()), "init not deterministic"
        zero_dropout(model2)
        N = 3 * B + 7                                                             # last batch is partial
        cols2, y2 = synth_inputs(cfg, np.random.RandomState(7), N)
        model2.compile(torch.optim.Adam(model2.parameters(), lr=cfg['lr']), "binary_crossentropy",
                       metrics=["binary_crossentropy", "auc"])
        feed = {n: cols2[n] for n in names}
        hist = model2.fit(x=dict(feed), y=y2, batch_size=B, epochs=2, verbose=0, shuffle=False)
        pred = model2.predict(dict(feed), batch_size=2 * B)
        for n in names:
            out[f"fit/x/{n}"] = np.asarray(cols2[n])
        out["fit/y"] = y2
        out["fit/loss"] = np.asarray(hist.history["loss"], dtype=np.float64)
        out["fit/pred"] = pred
        out["fit/pred_dtype"] = np.array(str(pred.dtype))
        out["fit/batch_size"] = np.array(B)

    meta = dict(name=name, fields=cfg['fields'], vocab=[vocab[f] for f in cfg['fields']], dense=cfg['dense'],
                domain=cfg['domain'], num_domains_list=[int(v) for v in num_domains_list], D=cfg['D'], H=cfg['H'],
                L=cfg['L'], units=list(cfg['units']), flag=cfg['flag'], mode=cfg['mode'], lr=cfg['lr'],
                seed=cfg['seed'], adam_steps=cfg['adam_steps'] if cfg['train'] and cfg.get('save_adam', True) else 0,
                feature_names=names,
                torch=torch.__version__)
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(outdir, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    _, all_cases = make_case('aliccp_sota')
    for case in (sys.argv[1:] or all_cases):
        run_case(case, outdir)
