import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sa_gnn_amd import ops
from sa_gnn_amd.model import random_fusion_params
dev = torch.device("cuda:0")
for d, t, n in [(64, 16, 10_000_000), (64, 16, 3_000_000), (64, 16, 5_000_000)]:
    g = torch.Generator(device=dev).manual_seed(d + t)
    x = torch.rand((t, n, d), generator=g, device=dev).mul_(2).sub_(1).permute(1, 0, 2)
    p = random_fusion_params(d, dev, 7)
    h = torch.empty((n, t, d), device=dev)
    f = lambda: ops.lstm_fwd(x, p["lstm_W"], p["lstm_b"], out=h)
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{sys.argv[1]} lstm d{d} t{t} n{n}: {np.median(ts):.2f} ms (min {min(ts):.2f}) per M rows {np.median(ts) / n * 1e6:.3f}", flush=True)
    del x, h
