import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from sa_gnn_amd import _lib, ops, autograd as ag
from oracle import selfgnn_oracle as O
lib = _lib.load(); dev = torch.device("cuda:0")
for d, t, n in [(32, 3, 33), (64, 2, 33), (32, 1, 2), (64, 2, 1537), (64, 3, 70001)]:
    rng = np.random.default_rng(d + t)
    x = rng.standard_normal((n, t, d)).astype(np.float32)
    p = O.init_fusion_params(d, rng)
    gout = rng.standard_normal((n, d)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    pd = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in p.items()}
    ops.range_redo_count(reset=True)
    got = ag.interval_fusion(xd, pd, 16)
    r_fwd = ops.range_redo_count(reset=True)
    # backward pieces by hand
    h = torch.empty((n, t, d), device=dev); gates = torch.empty((n, t, 4 * d), device=dev); cell = torch.empty((n, t, d), device=dev)
    ops.check(lib.sagnn_lstm_fwd_train_f32(xd.data_ptr(), t * d, d, n, t, d, pd["lstm_W"].data_ptr(), pd["lstm_b"].data_ptr(), 1.0, None, h.data_ptr(), t * d, gates.data_ptr(), cell.data_ptr(), None))
    r1 = ops.range_redo_count(reset=True)
    y2, qkv = ag._attn_bwd_front(h, pd["ln_gamma"].detach(), pd["ln_beta"].detach(), pd["Wq"], pd["bq"], pd["Wk"], pd["bk"], pd["Wv"], pd["bv"], 16, torch.from_numpy(gout).to(dev))
    r2 = ops.range_redo_count(reset=True)
    Wqkv = torch.cat([pd["Wq"], pd["Wk"], pd["Wv"]], dim=1).detach().contiguous()
    dW = torch.zeros((d, 3 * d), device=dev); db = torch.zeros(3 * d, device=dev)
    ops.check(lib.sagnn_attn_bwd_tail_f32(y2.data_ptr(), qkv.data_ptr(), n * t, d, Wqkv.data_ptr(), dW.data_ptr(), db.data_ptr(), None))
    r3 = ops.range_redo_count(reset=True)
    print(f"d{d} t{t} n{n}: redo fwd(train) {r_fwd}, lstm_train {r1}, front {r2}, tail {r3}; |qkv| rows min max {float(qkv.abs().max(1).values.min()):.3e}", flush=True)
