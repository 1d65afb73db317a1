import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import selfgnn_oracle as O
from sa_gnn_amd import ops
dev = torch.device("cuda:0")
for d, t, n in [(32, 2, 200003), (32, 16, 70001), (64, 16, 70001), (128, 6, 70001), (64, 3, 100003)]:
    rng = np.random.default_rng(d * 100 + t + 1)
    x = rng.standard_normal((n, t, d)).astype(np.float32)
    p = O.init_fusion_params(d, rng)
    pd = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}
    xd = torch.from_numpy(x).to(dev)
    outs = []
    for rep in range(3):
        outs.append(ops.ln_mhsa_mean(xd, pd["ln_gamma"], pd["ln_beta"], pd["Wq"], pd["bq"], pd["Wk"], pd["bk"], pd["Wv"], pd["bv"], 16).cpu().numpy())
    y = O.layer_norm_td(x, p["ln_gamma"], p["ln_beta"])
    want = O.mhsa(y, p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], 16).mean(axis=1)
    for rep in range(3):
        err = np.abs(outs[rep] - want)
        bad = err > 2e-5 + 1e-4 * np.abs(want)
        rows = np.nonzero(bad.any(axis=1))[0]
        print(f"d{d} t{t} rep{rep}: max err {err.max():.3e} bad rows {len(rows)} first {rows[:10].tolist()} cols {np.nonzero(bad.any(axis=0))[0][:16].tolist()} same as rep0 {np.array_equal(outs[rep], outs[0])} nan {int(np.isnan(outs[rep]).sum())}")
