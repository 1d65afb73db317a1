"""Is the round-wise, row-chunked fusion (parallel.RoundFusion) still bit-identical to one call? Prints redo counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sa_gnn_amd import ops
from sa_gnn_amd.model import random_fusion_params
dev = torch.device("cuda:0")
n, T, d = 10000, 16, 64
g = torch.Generator(device=dev).manual_seed(3)
xs = (torch.rand((T, n, d), generator=g, device=dev) * 0.06 - 0.03)
p = random_fusion_params(d, dev, 8)
ops.range_redo_count(True)
x = xs.permute(1, 0, 2)
h = ops.lstm_fwd(x, p["lstm_W"], p["lstm_b"])
print("full lstm redo", ops.range_redo_count(True))
f = ops.ln_mhsa_mean(h, p["ln_gamma"], p["ln_beta"], p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], 16)
print("full attn redo", ops.range_redo_count(True))
world = 4
rows = n // world
for r in range(world):
    xr = xs[:, r * rows:(r + 1) * rows, :].contiguous()          # [T, rows, d] as the exchange delivers it
    h2 = torch.empty((rows, T, d), device=dev)
    c = torch.empty((rows, d), device=dev)
    for lo, hi in ((0, rows // 2), (rows // 2, rows)):
        for j in range(4):
            sl = xr[4 * j:4 * j + 4, lo:hi, :].permute(1, 0, 2)
            ops.lstm_fwd(sl, p["lstm_W"], p["lstm_b"], out=h2[lo:hi, 4 * j:4 * j + 4, :], h0=h2[lo:hi, 4 * j - 1, :] if j else None,
                         c0=c[lo:hi] if j else None, c_out=c[lo:hi] if j < 3 else None)
        print(f"rank {r} rows {lo}:{hi} lstm redo", ops.range_redo_count(True), "h identical", bool(torch.equal(h2[lo:hi], h[r * rows + lo:r * rows + hi])))
        f2 = ops.ln_mhsa_mean(h2[lo:hi], p["ln_gamma"], p["ln_beta"], p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], 16)
        ref = f[r * rows + lo:r * rows + hi]
        bad = (f2 != ref).any(dim=1)
        print(f"   attn redo", ops.range_redo_count(True), "f identical", bool(torch.equal(f2, ref)), "rows differing", int(bad.sum()),
              "first", (torch.nonzero(bad)[:5].flatten().tolist()), "max diff", float((f2 - ref).abs().max()))
