"""Which h rows of the bench's reduced-scale workload hold 4-element segments that are non-zero but below 2^-18?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sa_gnn_amd import ops, synthetic
from sa_gnn_amd.model import random_fusion_params
dev = torch.device("cuda:0")
scale = 0.002
U, I, d, L, T = int(10_000_000 * scale), int(5_000_000 * scale), 64, 2, 16
out_u = torch.empty((T, U, d), device=dev); out_i = torch.empty((T, I, d), device=dev)
scr_u = torch.empty((2, U, d), device=dev); scr_i = torch.empty((2, I, d), device=dev)
for k in range(T):
    u, i = synthetic.powerlaw_edges(U, I, int(100_000_000 * scale), seed=1000 + k, device=dev, zipf_s=0.8)
    (rp_u, ci_u), (rp_i, ci_i) = synthetic.csr_pair_from_edges(u, i, U, I)
    pu = ops.SpmmPlan(rp_u, ci_u, U, I, device=dev, validate=False); pi = ops.SpmmPlan(rp_i, ci_i, I, U, device=dev, validate=False)
    g = torch.Generator(device=dev); g.manual_seed(2000 + k)
    u0 = (torch.rand((U, d), generator=g, device=dev) * 0.02 - 0.01); i0 = (torch.rand((I, d), generator=g, device=dev) * 0.02 - 0.01)
    ops.gnn_interval(pu, pi, u0, i0, L, 0.5, out_u[k], out_i[k], scr_u, scr_i)
prm = [random_fusion_params(d, dev, seed) for seed in (7, 8)]
prm[1]["lstm_W"], prm[1]["lstm_b"] = prm[0]["lstm_W"], prm[0]["lstm_b"]
for tag, xs, p in (("users", out_u, prm[0]), ("items", out_i, prm[1])):
    x = xs.permute(1, 0, 2)
    ops.range_redo_count(True)
    h = ops.lstm_fwd(x, p["lstm_W"], p["lstm_b"])
    print(tag, "one call redo", ops.range_redo_count(True), "|x| min row max", float(x.abs().amax(dim=(1, 2)).min()), "|h| stats", float(h.abs().mean()), float(h.abs().max()))
    seg = h.view(h.shape[0], T, d // 4, 4).abs().amax(-1)
    tiny = (seg > 0) & (seg < 2.0 ** -18)
    print("   tiny h segments:", int(tiny.sum()), "rows", torch.nonzero(tiny.any(-1).any(-1))[:8].flatten().tolist())
    segx = x.reshape(x.shape[0], T, d // 4, 4).abs().amax(-1)
    tx = (segx > 0) & (segx < 2.0 ** -18)
    print("   tiny x segments:", int(tx.sum()))
    if tiny.any():
        r, t_, s_ = torch.nonzero(tiny)[0].tolist()
        print("   e.g. row", r, "step", t_, "segment", s_, h[r, t_, 4 * s_:4 * s_ + 4].tolist(), "x row max", float(x[r].abs().max()), "h row", h[r, t_, :8].tolist())
