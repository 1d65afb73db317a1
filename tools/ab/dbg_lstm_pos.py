import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sa_gnn_amd import ops
from sa_gnn_amd.model import random_fusion_params
dev = torch.device("cuda:0")
n, T, d = 960, 4, 64
g = torch.Generator(device=dev).manual_seed(3)
x = (torch.rand((n, T, d), generator=g, device=dev) * 0.06 - 0.03)
p = random_fusion_params(d, dev, 8)
h = ops.lstm_fwd(x, p["lstm_W"], p["lstm_b"])
for off in (0, 2, 16, 30, 96):
    h2 = ops.lstm_fwd(x[off:].contiguous(), p["lstm_W"], p["lstm_b"])
    diff = (h2 != h[off:])
    idx = torch.nonzero(diff)
    print("offset", off, "differing elements", int(diff.sum()), "rows", sorted(set(idx[:, 0].tolist()))[:24], "steps", sorted(set(idx[:, 1].tolist())), "cols", sorted(set(idx[:, 2].tolist()))[:20])
    if len(idx):
        r, t, c = idx[0].tolist()
        print("   e.g.", (r, t, c), float(h2[r, t, c]), float(h[off + r, t, c]), "max abs diff", float((h2 - h[off:]).abs().max()))
