"""A/B of the one-launch BPTT (dW on the fp32 MFMA inside the launch) against BPTT-without-dW + the f16 x 2 weight-gradient pass
(sagnn_lstm_bwd_ws_f32): same process, same operands. Prints times and the dW / dx / db differences."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sa_gnn_amd import autograd as ag, ops
from sa_gnn_amd.model import random_fusion_params
dev = torch.device("cuda:0")
lib = ops._lib.load()
for d, t, n in [(64, 3, 70_001), (32, 4, 50_003), (64, 2, 8_000_000), (64, 16, 1_000_000), (32, 6, 4_000_000)]:
    g = torch.Generator(device=dev).manual_seed(d + t)
    x = torch.rand((t, n, d), generator=g, device=dev).mul_(2).sub_(1).permute(1, 0, 2)
    p = random_fusion_params(d, dev, 7)
    h = torch.empty((n, t, d), device=dev); gates = torch.empty((n, t, 4 * d), device=dev); cell = torch.empty((n, t, d), device=dev)
    ops.check(lib.sagnn_lstm_fwd_train_f32(x.data_ptr(), x.stride(0), x.stride(1), n, t, d, p["lstm_W"].data_ptr(), p["lstm_b"].data_ptr(), 1.0, None,
                                           h.data_ptr(), t * d, gates.data_ptr(), cell.data_ptr(), None))
    dh = torch.randn((n, t, d), generator=g, device=dev) * torch.rand((n, 1, 1), generator=g, device=dev).mul(-12).exp2()   # rows of 1 .. 2^-12
    res = {}
    for mode in (False, True):
        ag.SPLIT_DW = mode
        f = lambda: ag.lstm_bwd(x, h, gates, cell, dh, None, p["lstm_W"])
        out = f(); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); out = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        res[mode] = (np.median(ts), out)
    (ta, a), (tb, b) = res[False], res[True]
    rel = lambda u, v: float((u - v).abs().max() / v.abs().max())
    print(f"d{d} t{t} n{n}: one-launch {ta:.2f} ms, split {tb:.2f} ms; dx equal {bool(torch.equal(a[0], b[0]))} dW rel {rel(b[1], a[1]):.2e} db rel {rel(b[2], a[2]):.2e} redo {ops.range_redo_count(reset=True)}", flush=True)
    del x, h, gates, cell, dh, res, a, b, out
