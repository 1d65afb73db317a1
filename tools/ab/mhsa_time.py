import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sa_gnn_amd import ops
from sa_gnn_amd.model import random_fusion_params
dev = torch.device("cuda:0")
for d, t, n in [(64, 16, 1_500_000), (64, 8, 3_000_000), (64, 2, 8_000_000), (64, 6, 3_000_000), (32, 16, 2_000_000)]:
    g = torch.Generator(device=dev).manual_seed(d + t)
    x = torch.rand((t, n, d), generator=g, device=dev).mul_(2).sub_(1).permute(1, 0, 2)
    p = random_fusion_params(d, dev, 7)
    f = lambda: ops.ln_mhsa_mean(x, p["ln_gamma"], p["ln_beta"], p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], 16)
    o = f(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); o = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{sys.argv[1]} mhsa d{d} t{t} n{n}: {np.median(ts):.2f} ms (min {min(ts):.2f})  checksum {float(o.double().abs().mean()):.12f}", flush=True)
    del x, o
