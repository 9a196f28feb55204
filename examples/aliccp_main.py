#!/usr/bin/env python3
"""The SATrans branch of the reference's main.py (main.py:95-133, 182-191, 292-306, 343-400) on this package: load the AliCCP
columns from `alicpp.h5` (datasets `ctr_train/<col>`, `ctr_test/<col>`, as aliccp_dataset_processing.py writes them), build the
feature columns with the reference's vocabulary sizes, train, predict, report the overall and per-scenario AUC and the test
loss, append the result line the reference appends to `<data>_results.csv`, optionally dump the state_dict.

    python examples/aliccp_main.py --h5 /data/alicpp.h5 --domain_col 301 --flag sota --embedding_dim 32 --att_layer_num 3 \\
        --att_head_num 4 --meta_mode QK --learning_rate 0.005 --seed 1021 --batch_size 8192 --epochs 1
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 examples/aliccp_main.py --h5 ... (one process per GPU, row ownership)

Differences from the reference script: the two import lines, the HDF5 columns come through
satrans_amd.pipeline.load_h5_columns (no h5py needed, memory-mapped), and the evaluation report is one call
(`evaluate_domains`: the same numbers as main.py:353-374, computed on the device).  `--data_max` overrides the reference's
hard-coded column maxima (main.py:124-127) for other datasets with the same layout."""
import argparse
import json
import os
import sys
from datetime import datetime

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from satrans_amd import SATrans, SparseFeat, get_feature_names  # noqa: E402
from satrans_amd.pipeline import load_h5_columns  # noqa: E402

SPARSE = ['101', '121', '122', '124', '125', '126', '127', '128', '129', '205', '206', '207', '210', '216', '508', '509', '702',
          '853', '301']                                                             # main.py:99-101
DATA_MAX = {'101': 444861, '121': 97, '122': 13, '124': 2, '125': 7, '126': 3, '127': 3, '128': 2, '129': 4, '205': 4348615,
            '206': 8993, '207': 695124, '210': 99606, '216': 234880, '508': 8185, '509': 472354, '702': 167813, '853': 91358,
            '301': 3}                                                               # main.py:124-127


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--h5", required=True)
    ap.add_argument("--postfix", default="")
    ap.add_argument("--domain_col", default="301")
    ap.add_argument("--flag", default="sota")
    ap.add_argument("--embedding_dim", type=int, default=32)
    ap.add_argument("--att_layer_num", type=int, default=3)
    ap.add_argument("--att_head_num", type=int, default=4)
    ap.add_argument("--meta_mode", default="QK")
    ap.add_argument("--learning_rate", type=float, default=0.005)
    ap.add_argument("--seed", default="1021")
    ap.add_argument("--batch_size", type=int, default=8192)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--data_max", default=None, help="JSON {column: max id} instead of the reference's AliCCP maxima")
    ap.add_argument("--results", default=None, help="CSV to append the reference's result line to (default: none)")
    ap.add_argument("--dump", default=None, help="path for torch.save(model.cpu().state_dict()) (reference flag 'dump')")
    args = ap.parse_args(argv)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:                                                                   # one process per GPU over RCCL
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    device = f"cuda:{local}"
    data_max = dict(DATA_MAX, **(json.loads(args.data_max) if args.data_max else {}))
    cols = ['click'] + SPARSE
    train = dict(load_h5_columns(args.h5, 'ctr_train' + args.postfix, cols))       # main.py:106-109 (get_aliccp_ctr_df)
    test = dict(load_h5_columns(args.h5, 'ctr_test' + args.postfix, cols))
    if int(np.min(train['301'])) == 0:                                              # scenario ids start at 1 (main.py:112-114)
        train['301'] = np.asarray(train['301']) + 1
        test['301'] = np.asarray(test['301']) + 1
    if world > 1:                                                                   # every rank its shard, equal sizes
        n = (len(train['click']) // world)
        train = {k: np.asarray(v)[rank * n:(rank + 1) * n] for k, v in train.items()}
    domain_cols = [args.domain_col]
    num_domains_list = [max(len(np.unique(train[c])), data_max[c]) for c in domain_cols]          # main.py:131-132
    columns = [SparseFeat(f, vocabulary_size=int(data_max[f]) + 2, embedding_dim=args.embedding_dim) for f in SPARSE]
    names = get_feature_names(columns)
    model = SATrans(columns, columns, domain_cols, num_domains_list, att_layer_num=0, domain_att_layer_num=args.att_layer_num,
                    att_head_num=args.att_head_num, use_linear=False, use_dnn=False, meta_mode=args.meta_mode, seed=args.seed,
                    device=device, flag=args.flag)                                   # main.py:292-306
    model.compile(torch.optim.Adam(model.parameters(), lr=args.learning_rate), "binary_crossentropy",
                  metrics=["binary_crossentropy", "auc"])                           # main.py:343
    x_train = {f: train[f] for f in names}
    model.fit(x=x_train, y=np.asarray(train['click']), batch_size=args.batch_size, epochs=args.epochs,
              verbose=1 if rank == 0 else 0)                                        # main.py:345-349
    x_test = {f: test[f] for f in names}
    rep = model.evaluate_domains(x_test, np.asarray(test['click']), args.batch_size * 4, domain_col=args.domain_col)
    aucs = [round(rep["auc"], 4)] + [round(rep["domain_auc"][i], 4) for i in sorted(rep["domain_auc"])]
    if rank == 0:
        print("test AUC", aucs[0])
        for i, a in zip(sorted(rep["domain_auc"]), aucs[1:]):
            print(f"Domain {i} test AUC", a)
        line = (f"{datetime.now().strftime('%m-%d-%H-%M')}-SATrans_{args.embedding_dim}_{args.learning_rate}_{args.att_layer_num}_"
                f"{args.att_head_num}_{args.meta_mode}_{args.seed}_{args.domain_col}_{args.flag}," +
                ",".join(str(a) for a in aucs) + "," + "%.6f" % rep["loss"])       # main.py:387-395
        print(line)
        if args.results:
            with open(args.results, "a") as f:
                f.write(line + "\n")
        if args.dump:
            torch.save(model.cpu().state_dict(), args.dump)                         # main.py:399-400
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return rep


if __name__ == "__main__":
    main()
