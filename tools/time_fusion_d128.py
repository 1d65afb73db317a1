import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import selfgnn_oracle as O
from sa_gnn_amd import ops
from sa_gnn_amd.model import random_fusion_params
dev = torch.device("cuda:0")
d, t = 128, 6
for n in (33_000, 1_000_000):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand((t, n, d), generator=g, device=dev).mul_(2).sub_(1).permute(1, 0, 2)
    p = random_fusion_params(d, dev, 7)
    S = min(n, 20000)
    pn = {k: v.cpu().numpy() for k, v in p.items()}
    want_h = O.basic_lstm(np.ascontiguousarray(x[:S].cpu().numpy()), pn["lstm_W"], pn["lstm_b"], 1.0)
    want_f = O.interval_fusion(np.ascontiguousarray(x[:S].cpu().numpy()), pn, 16)
    for mode in ("split", "valu"):
        ops.set_engine("valu" if mode == "valu" else "f16x2")
        def timed(fn, reps=5):
            fn(); torch.cuda.synchronize(); ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ts))
        h = torch.empty((n, t, d), device=dev)
        ms = timed(lambda: ops.lstm_fwd(x, p["lstm_W"], p["lstm_b"], out=h))
        err = np.abs(h[:S].cpu().numpy() - want_h).max()
        out = [None]
        def fuse(): out[0] = ops.interval_fusion(x, p, 16)
        msf = timed(fuse)
        errf = np.abs(out[0][:S].cpu().numpy() - want_f).max()
        print(f"[{mode}] d={d} t={t} n={n}: LSTM {ms:.3f} ms (h err {err:.2e}); whole fusion {msf:.3f} ms (fused err {errf:.2e})", flush=True)
