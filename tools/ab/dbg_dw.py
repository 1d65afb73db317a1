import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from sa_gnn_amd import _lib, ops
from sa_gnn_amd.model import random_fusion_params
import test_gpu_f16_range as tr
lib = _lib.load(); dev = torch.device("cuda:0")
d, t, n, pattern = 64, 2, 2, "blocks"
rng = np.random.default_rng(d + t + len(pattern))
g = torch.Generator(device="cpu").manual_seed(d * 3 + t)
x = (torch.rand((n, t, d), generator=g) * 2 - 1).to(dev)
p = random_fusion_params(d, dev, 11)
h = torch.empty((n, t, d), device=dev); gates = torch.empty((n, t, 4 * d), device=dev); cell = torch.empty((n, t, d), device=dev)
ops.check(lib.sagnn_lstm_fwd_train_f32(x.data_ptr(), t * d, d, n, t, d, p["lstm_W"].data_ptr(), p["lstm_b"].data_ptr(), 1.0, None, h.data_ptr(), t * d, gates.data_ptr(), cell.data_ptr(), None))
scale = torch.from_numpy(tr._row_scales(n, rng, pattern).astype(np.float32)).to(dev)
print("scales", scale.tolist())
dh = torch.randn((n, t, d), generator=g).to(dev) * scale[:, None, None]
dx = torch.empty((n, t, d), device=dev); dW = torch.zeros((2 * d, 4 * d), device=dev); db = torch.zeros(4 * d, device=dev)
nbytes = int(lib.sagnn_lstm_bwd_workspace_bytes(n, t, d)); ws = torch.zeros(nbytes // 4, device=dev)
ops.range_redo_count(reset=True)
ops.check(lib.sagnn_lstm_bwd_ws_f32(x.data_ptr(), t * d, d, h.data_ptr(), gates.data_ptr(), cell.data_ptr(), dh.data_ptr(), t * d, None, p["lstm_W"].data_ptr(), dx.data_ptr(), dW.data_ptr(), db.data_ptr(), n, t, d, ws.data_ptr(), nbytes, None))
print("redo", ops.range_redo_count())
dG = ws.view(t, n, 4 * d).double()
xh = torch.cat([x.permute(1, 0, 2), torch.cat([torch.zeros((1, n, d), device=dev), h.permute(1, 0, 2)[:-1]])], dim=2).double()
want = torch.einsum("tnk,tng->kg", xh, dG); mag = torch.einsum("tnk,tng->kg", xh.abs(), dG.abs())
err = (dW.double() - want).abs(); rel = err / (mag + 1e-300)
k, gg = divmod(int(rel.argmax()), 4 * d)
print("worst", float(rel.max()), "at", k, gg, "got", float(dW[k, gg]), "want", float(want[k, gg]), "mag", float(mag[k, gg]))
for s in range(t):
    for i in range(n):
        print(" row", s, i, "xh", float(xh[s, i, k]), "dG", float(dG[s, i, gg]), "rowmax|dG|", float(dG[s, i].abs().max()), "rowmax|xh|", float(xh[s, i].abs().max()))
print("count rel>5e-7:", int((rel > 5e-7).sum()), "of", rel.numel())
