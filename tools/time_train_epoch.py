#!/usr/bin/env python3
"""Times Recommender.trainEpoch / testEpoch on a Gowalla-shaped synthetic dataset with the reference's
gowalla.sh hyper-parameters (U = 48,653, I = 52,619, 3 intervals x 600 k edges, d = 64, batch 512,
trnNum 10000 -> 20 steps per epoch, keepRate 0.5) and prints where a training step spends its time
(host sampling / forward + loss / backward / optimiser), wall clock with a device sync after each part."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sa_gnn_amd import Params, synthetic      # noqa: E402
from sa_gnn_amd.DataHandler import DataHandler   # noqa: E402
from sa_gnn_amd.Params import args            # noqa: E402
from sa_gnn_amd.Utils import NNLayers as NNs  # noqa: E402
from sa_gnn_amd.model import Recommender      # noqa: E402


def main():
    Params.parse_args("--data gowalla --lr 2e-3 --reg 1e-2 --ssl_reg 1e-6 --epoch 150 --batch 512 --sslNum 40 --graphNum 3 "
                      "--gnn_layer 2 --att_layer 1 --testSize 1000 --ssldim 48 --keepRate 0.5".split(), namespace=args)
    np.random.seed(100)
    U, I = 48653, 52619
    tmt = synthetic.make_trn_mat_time(U, I, [600000] * 3)
    seq = synthetic.make_sequence(tmt)
    rng = np.random.default_rng(1)
    tst = [None] * U
    for u in rng.choice(U, 10000, replace=False):
        tst[u] = int(rng.integers(0, I))
    test_dict = {u + 1: list(rng.integers(1, I + 1, size=1000)) for u in range(U)}
    h = DataHandler.from_memory(tmt, seq, tst, test_dict)
    rec = Recommender(torch.device("cuda:0"), h)
    rec.prepareModel()
    for _ in range(2):
        rec.trainEpoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_ep = 5
    for _ in range(n_ep):
        rec.trainEpoch()
    torch.cuda.synchronize()
    ep = (time.perf_counter() - t0) / n_ep
    steps = int(np.ceil(args.trnNum / args.batch))
    print(f"train epoch {ep * 1e3:.1f} ms = {steps} steps of {ep / steps * 1e3:.2f} ms")
    # one step, by part
    parts = {"sample": 0.0, "forward+loss": 0.0, "backward": 0.0, "optimiser": 0.0}
    sf = np.random.permutation(args.user)[:args.trnNum]
    for i in range(steps):
        bat = sf[i * args.batch:(i + 1) * args.batch]
        torch.cuda.synchronize(); t = time.perf_counter()
        uL, iL, sq, mk, uLs = rec.sampleTrainBatch(bat, h.trnMat, h.timeMat, 40, as_arrays=True)
        su, si, _ = rec.sampleSslBatch(bat, h.subMat, False, as_arrays=True)
        batch = {"uids": uL, "iids": iL, "uLocs_seq": uLs, "sequence": sq, "mask": mk, "suids": su, "siids": si}
        parts["sample"] += time.perf_counter() - t; t = time.perf_counter()
        params = rec._trainable()
        for p in params.values():
            p.grad = None
        pre, ssl = rec.train_loss(batch)
        loss = pre + args.ssl_reg * ssl
        torch.cuda.synchronize(); parts["forward+loss"] += time.perf_counter() - t; t = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); parts["backward"] += time.perf_counter() - t; t = time.perf_counter()
        rec.optimizer.step({k: p.grad for k, p in params.items()})
        torch.cuda.synchronize(); parts["optimiser"] += time.perf_counter() - t
    print("per step (ms, synced between parts):", {k: round(v / steps * 1e3, 3) for k, v in parts.items()})
    t0 = time.perf_counter()
    res = rec.testEpoch()
    torch.cuda.synchronize()
    print(f"test epoch ({len(h.tstUsrs)} users) {1e3 * (time.perf_counter() - t0):.1f} ms", {k: round(v, 4) for k, v in res.items()})


if __name__ == "__main__":
    main()
