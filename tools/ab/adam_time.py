import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sa_gnn_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
params = {"a": torch.randn(640_000_000, generator=g, device=dev), "b": torch.randn(320_000_001, generator=g, device=dev), "c": torch.randn(4099, generator=g, device=dev)}
grads = {k: torch.randn(v.shape, generator=g, device=dev) for k, v in params.items()}
opt = ops.Adam(params, lr=1e-3, decay=0.96, decay_step=19, reg=1e-2, reg_names=["a", "b"])
opt.step(grads); torch.cuda.synchronize()
ts = []
for _ in range(6):
    t0 = time.perf_counter(); opt.step(grads); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
n = sum(v.numel() for v in params.values())
print(f"{sys.argv[1]} adam {n/1e6:.0f} M elements: {np.median(ts):.3f} ms (min {min(ts):.3f}) -> {7 * 4 * n / np.median(ts) / 1e9:.2f} TB/s; checksum {float(sum(v.double().abs().sum() for v in params.values())):.6f}")
