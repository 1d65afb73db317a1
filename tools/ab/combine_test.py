"""A/B of the attention kernel's cross-lane combine (SAGNN_ATTN_COMBINE = 0 LDS table / 1 ds_bpermute / 2 DPP).
usage: SAGNN_LIB=<lib> python tools/ab/combine_test.py {ref|check} tag"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sa_gnn_amd import ops
from sa_gnn_amd.model import random_fusion_params
mode, tag = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
out = {}
for d, t, n in [(64, 16, 1_500_000), (64, 8, 2_000_000), (64, 12, 1_500_000), (32, 16, 1_000_000), (32, 8, 1_000_000), (128, 8, 400_000), (128, 12, 300_000), (128, 16, 300_000), (128, 6, 400_000)]:
    g = torch.Generator(device=dev).manual_seed(d + t)
    x = torch.rand((n, t, d), generator=g, device=dev).mul_(2).sub_(1)
    p = random_fusion_params(d, dev, 7)
    f = lambda: ops.ln_mhsa_mean(x, p["ln_gamma"], p["ln_beta"], p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], 16)
    y = f(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); y2 = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    same = bool(torch.equal(y, y2))
    key = f"d{d}_t{t}"
    if mode == "ref":
        torch.save(y.cpu(), f"/tmp/ref_{key}.pt")
        print(f"{tag} {key}: {np.median(ts):.2f} ms (min {min(ts):.2f}); run-to-run identical {same}", flush=True)
    else:
        ref = torch.load(f"/tmp/ref_{key}.pt").to(dev)
        err = (y - ref).abs()
        bad_rows = int((err.amax(dim=1) > 1e-5 * (1 + ref.abs().amax(dim=1))).sum())
        print(f"{tag} {key}: {np.median(ts):.2f} ms (min {min(ts):.2f}); max |diff| vs LDS form {float(err.max()):.3e}; rows off {bad_rows} of {n}; run-to-run identical {same}", flush=True)
    del x, y, y2
