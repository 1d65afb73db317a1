"""Per-interval SpMM stack of the 100 M-edge graphs: one launch per layer AND direction (sagnn_gnn_interval_f32, what bench.py
times) against one launch per layer for both directions (sagnn_gnn_stack_f32 on a one-interval batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sa_gnn_amd import ops, synthetic
dev = torch.device("cuda:0")
U, I, d, L, T = 10_000_000, 5_000_000, 64, 2, 3
plans, emb, batches = [], [], []
for k in range(T):
    u, i = synthetic.powerlaw_edges(U, I, 100_000_000, seed=1000 + k, device=dev)
    (rp_u, ci_u), (rp_i, ci_i) = synthetic.csr_pair_from_edges(u, i, U, I)
    del u, i
    pu, pi = ops.SpmmPlan(rp_u, ci_u, U, I, device=dev, validate=False), ops.SpmmPlan(rp_i, ci_i, I, U, device=dev, validate=False)
    plans.append((pu, pi)); batches.append(ops.SpmmBatch([pu], [pi]))
    g = torch.Generator(device=dev); g.manual_seed(2000 + k)
    emb.append((torch.rand((1, U, d), generator=g, device=dev) * 0.02 - 0.01, torch.rand((1, I, d), generator=g, device=dev) * 0.02 - 0.01))
out_u, out_i = torch.empty((T, U, d), device=dev), torch.empty((T, I, d), device=dev)
out_u2, out_i2 = torch.empty((T, U, d), device=dev), torch.empty((T, I, d), device=dev)
scr_u, scr_i = torch.empty((2, U, d), device=dev), torch.empty((2, I, d), device=dev)
sb_u, sb_i = torch.empty((2, 1, U, d), device=dev), torch.empty((2, 1, I, d), device=dev)

def per_dir():
    for k in range(T):
        ops.gnn_interval(plans[k][0], plans[k][1], emb[k][0][0], emb[k][1][0], L, 0.5, out_u[k], out_i[k], scr_u, scr_i)

def both_dirs():
    for k in range(T):
        ops.gnn_stack(batches[k], emb[k][0], emb[k][1], L, 0.5, out_u2[k:k + 1], out_i2[k:k + 1], sb_u, sb_i)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))

for _ in range(2):
    a, b = timed(per_dir), timed(both_dirs)
    print(f"{T} intervals x {L} layers: per direction {a:.2f} ms, both directions per launch {b:.2f} ms; equal {bool(torch.equal(out_u, out_u2) and torch.equal(out_i, out_i2))}", flush=True)
