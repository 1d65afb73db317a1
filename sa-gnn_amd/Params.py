"""Flag system with the reference's names and defaults (reference Params.py:3-53).

Differences, all host-side: flags are declared from one table; `args` is built from defaults at
import and main.py applies the command line through `parse_args(argv)` — the reference parses
sys.argv at import time (Params.py:52), which breaks any importer that owns argv (pytest,
torchrun). Flags the reference never reads are still accepted so its *.sh lines run unchanged.
"""
from __future__ import annotations

import argparse

# name, type, default, help
_FLAGS = [
    ("lr", float, 1e-3, "learning rate"),
    ("batch", int, 512, "batch size"),
    ("testbatch", int, 64, "unused by the reference"),
    ("reg", float, 1e-5, "weight decay regularizer"),
    ("epoch", int, 100, "number of epochs"),
    ("graphNum", int, 8, "T: number of time-interval graphs"),
    ("decay", float, 0.96, "learning-rate decay"),
    ("save_path", str, "tem", "checkpoint / history name"),
    ("latdim", int, 64, "d: embedding size"),
    ("ssldim", int, 32, "SSL meta-net width"),
    ("rank", int, 4, "unused"),
    ("memosize", int, 2, "unused"),
    ("sampNum", int, 40, "unused"),
    ("testSize", int, 100, "candidates per test user"),
    ("sslNum", int, 20, "SSL pairs per user"),
    ("query_vector_dim", int, 64, "AdditiveAttention width (constructed, never applied)"),
    ("num_attention_heads", int, 16, "MHSA heads"),
    ("hyperNum", int, 128, "unused"),
    ("gnn_layer", int, 2, "L: GNN layers per interval"),
    ("trnNum", int, 10000, "training users per epoch"),
    ("load_model", str, None, "checkpoint to resume"),
    ("shoot", int, 10, "K of top-K"),
    ("data", str, "yelp", "dataset directory name"),
    ("target", str, "buy", "unused"),
    ("deep_layer", int, 0, "unused"),
    ("mult", float, 100, "unused"),
    ("keepRate", float, 0.5, "dropout keep probability"),
    ("slot", float, 1, "dead code only"),
    ("graphSampleN", int, 15000, "dead code only"),
    ("divSize", int, 10000, "unused"),
    ("tstEpoch", int, 3, "test every N epochs"),
    ("subUsrSize", int, 10, "unused"),
    ("subUsrDcy", float, 0.9, "unused"),
    ("leaky", float, 0.5, "leaky-ReLU slope"),
    ("hyperReg", float, 1e-4, "unused"),
    ("temp", float, 1, "unused"),
    ("ssl_reg", float, 1e-4, "SSL loss weight"),
    ("percent", float, 0.0, "noise percentage"),
    ("pos_length", int, 200, "max sequence length"),
    ("att_size", int, 12000, "unused"),
    ("att_layer", int, 4, "sequence attention layers"),
    ("pred_num", int, 5, "prediction targets sampled in training"),
    ("nfs", bool, False, "unused"),
    ("test", bool, True, "test (True) or validation"),
    ("ssl", bool, True, "unused"),
    ("uid", int, 0, "debug print index"),
]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Model Params")
    for name, typ, default, text in _FLAGS:
        # type=bool keeps the reference's semantics: any non-empty string is True (Params.py:47-49)
        p.add_argument("--" + name, default=default, type=typ, help=text)
    return p


def parse_args(argv=None, namespace=None):
    ns = build_parser().parse_args([] if argv is None else argv, namespace)
    ns.decay_step = ns.trnNum // ns.batch          # reference Params.py:53
    return ns


args = parse_args([])
# args.user / args.item are injected by DataHandler.LoadData (reference DataHandler.py:126)
