#!/usr/bin/env python3
"""Entry script with the reference's shape (reference main.py:12-26):
    python main.py --data gowalla --graphNum 3 --gnn_layer 2 --latdim 64 ...   (the *.sh lines work unchanged)
Seeds everything with 100 (main.py:21-23), loads Datasets/<data>/ through DataHandler and runs
Recommender.run() on cuda:0. `--data synthetic` builds a small in-memory dataset in the
reference's format instead (no dataset blob ships with the reference)."""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from sa_gnn_amd import Params
    Params.parse_args(sys.argv[1:], namespace=Params.args)        # fills the module-global args in place
    from sa_gnn_amd.DataHandler import DataHandler
    from sa_gnn_amd.Params import args
    from sa_gnn_amd.model import Recommender

    np.random.seed(100)
    random.seed(100)
    torch.manual_seed(100)
    if args.data == "synthetic":
        from sa_gnn_amd import synthetic
        U, I, T = 2000, 1500, args.graphNum
        tmt = synthetic.make_trn_mat_time(U, I, [30000] * T)
        seq = synthetic.make_sequence(tmt)
        rng = np.random.default_rng(100)
        tst = [int(rng.integers(0, I)) if u % 2 == 0 else None for u in range(U)]
        tdict = {u + 1: list(rng.integers(1, I + 1, size=args.testSize)) for u in range(U)}
        handler = DataHandler.from_memory(tmt, seq, tst, tdict)
    else:
        handler = DataHandler()
        handler.LoadData()
    print("Load Data")
    rec = Recommender(torch.device("cuda:0"), handler)
    rec.run()
