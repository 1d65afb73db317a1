"""Can the interval LSTM of the intervals that are done run BESIDE the SpMMs of the later ones (second stream, capped grid)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sa_gnn_amd import _lib, ops, synthetic
from sa_gnn_amd.model import random_fusion_params
dev = torch.device("cuda:0")
lib = _lib.load()
U, I, d, L, T = 10_000_000, 5_000_000, 64, 2, 4
plans, emb = [], []
for k in range(T):
    u, i = synthetic.powerlaw_edges(U, I, 100_000_000, seed=1000 + k, device=dev)
    (rp_u, ci_u), (rp_i, ci_i) = synthetic.csr_pair_from_edges(u, i, U, I)
    del u, i
    plans.append((ops.SpmmPlan(rp_u, ci_u, U, I, device=dev, validate=False), ops.SpmmPlan(rp_i, ci_i, I, U, device=dev, validate=False)))
    g = torch.Generator(device=dev); g.manual_seed(2000 + k)
    emb.append((torch.rand((U, d), generator=g, device=dev) * 0.02 - 0.01, torch.rand((I, d), generator=g, device=dev) * 0.02 - 0.01))
torch.cuda.empty_cache()
p = random_fusion_params(d, dev, 7)
out_u, out_i = torch.empty((T, U, d), device=dev), torch.empty((T, I, d), device=dev)
scr_u, scr_i = torch.empty((2, U, d), device=dev), torch.empty((2, I, d), device=dev)
h_u, h_i = torch.empty((U, T, d), device=dev), torch.empty((I, T, d), device=dev)
c_u, c_i = torch.empty((U, d), device=dev), torch.empty((I, d), device=dev)
side = torch.cuda.Stream()

def spmm(k):
    ops.gnn_interval(plans[k][0], plans[k][1], emb[k][0], emb[k][1], L, 0.5, out_u[k], out_i[k], scr_u, scr_i)

def lstm_step(k):
    for xs, h, c in ((out_u, h_u, c_u), (out_i, h_i, c_i)):
        ops.lstm_fwd(xs[k:k + 1].permute(1, 0, 2), p["lstm_W"], p["lstm_b"], out=h[:, k:k + 1, :], h0=h[:, k - 1, :] if k else None,
                     c0=c if k else None, c_out=c)

def serial():
    for k in range(T):
        spmm(k)
    for xs, h in ((out_u, h_u), (out_i, h_i)):
        ops.lstm_fwd(xs.permute(1, 0, 2), p["lstm_W"], p["lstm_b"], out=h)

def overlapped(limit):
    main = torch.cuda.current_stream()
    evs = []
    for k in range(T):
        spmm(k)
        ev = torch.cuda.Event(); ev.record(main); evs.append(ev)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            lib.sagnn_set_grid_limit(limit)
            lstm_step(k)
            lib.sagnn_set_grid_limit(0)
    main.wait_stream(side)

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))

t_spmm = timed(lambda: [spmm(k) for k in range(T)])
t_ser = timed(serial)
ref_u, ref_i = h_u.clone(), h_i.clone()
print(f"SpMM only {t_spmm:.1f} ms; serial SpMM + one LSTM call per node type {t_ser:.1f} ms (T = {T})", flush=True)
for limit in (0, 32, 64, 96, 128):
    t_ov = timed(lambda: overlapped(limit))
    same = bool(torch.equal(h_u, ref_u) and torch.equal(h_i, ref_i))
    print(f"overlapped, LSTM per interval on a second stream, grid limit {limit}: {t_ov:.1f} ms; h identical to the one-call LSTM {same}", flush=True)
