import sys, numpy as np, torch
sys.path.insert(0, '.')
from sa_gnn_amd import ops
from oracle import selfgnn_oracle as O
dev = torch.device('cuda:0')
rng = np.random.default_rng(9)
n_rows, n_src, d = 300, 64, 64
degs = rng.integers(0, 17, size=n_rows)
cols = [np.sort(rng.choice(n_src, size=int(dg), replace=False)) for dg in degs]
rowptr = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int32)
colidx = np.concatenate(cols).astype(np.int32)
idx = np.stack([np.repeat(np.arange(n_rows), np.diff(rowptr)), colidx], 1).astype(np.int32)
x = rng.standard_normal((n_src, d)).astype(np.float32)
plan = ops.SpmmPlan(rowptr, colidx, n_rows, n_src, device=dev)
y = ops.spmm(plan, torch.from_numpy(x).to(dev), 1.0).cpu().numpy()
want = O.message_propagate_zero_fill(x, idx, n_rows, 1.0)
bad = np.flatnonzero(np.abs(y - want).max(1) > 1e-4)
print("bad rows", bad)
print("deg", degs[bad])
print("lr", bad % 16)
for r in bad[:6]:
    # which subset of neighbours was summed? solve via least squares on x rows
    nb = cols[r]
    coef, *_ = np.linalg.lstsq(x[nb].T.astype(np.float64), y[r].astype(np.float64), rcond=None)
    print(r, degs[r], np.round(coef, 2))
