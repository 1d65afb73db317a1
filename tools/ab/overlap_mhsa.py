"""Can the users' LN+MHSA (issue-bound, HBM at 0.2) run BESIDE the item-side SpMMs of the last layer (HBM-bound, 12 % issuing)?
Serial sum against two streams, T = 16 fusion on 10 M nodes + 8 item-side SpMM launches of the roofline workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sa_gnn_amd import _lib, ops, synthetic
from sa_gnn_amd.model import random_fusion_params
dev = torch.device("cuda:0")
U, I, d, T, NS = 10_000_000, 5_000_000, 64, 16, 8
u, i = synthetic.powerlaw_edges(U, I, 100_000_000, seed=1000, device=dev)
(rp_u, ci_u), (rp_i, ci_i) = synthetic.csr_pair_from_edges(u, i, U, I)
del u, i, rp_u, ci_u
plan_i = ops.SpmmPlan(rp_i, ci_i, I, U, device=dev, validate=False)      # rows = items, gathers user rows
g = torch.Generator(device=dev); g.manual_seed(1)
xu = torch.rand((U, d), generator=g, device=dev) * 0.02 - 0.01
outs = torch.empty((NS, I, d), device=dev)
h = (torch.rand((T, U, d), generator=g, device=dev) * 2 - 1).permute(1, 0, 2)
p = random_fusion_params(d, dev, 7)
side = torch.cuda.Stream()
torch.cuda.empty_cache()

def spmms():
    for k in range(NS):
        ops.spmm(plan_i, xu, 0.5, out=outs[k])

def mhsa():
    return ops.ln_mhsa_mean(h, p["ln_gamma"], p["ln_beta"], p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], 16)

def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        mhsa()
    spmms()
    main.wait_stream(side)

def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))

a, b = timed(spmms), timed(mhsa)
c = timed(both)
print(f"{NS} item-side SpMMs {a:.2f} ms, users' LN+MHSA {b:.2f} ms, serial sum {a + b:.2f} ms; on two streams {c:.2f} ms ({(a + b) / c:.2f} x)", flush=True)
