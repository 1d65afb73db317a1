"""GPU parity of the backward pass of the interval stack against
torch.autograd over the oracle's torch restatement (CPU, float64)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import selfgnn_oracle as O

pytestmark = pytest.mark.gpu


def _case(rng, U, I, dens, d):
    m = sp.csr_matrix((rng.random((U, I)) < dens).astype(np.intc))
    u0 = rng.standard_normal((U, d)).astype(np.float32)
    i0 = rng.standard_normal((I, d)).astype(np.float32)
    gu = rng.standard_normal((U, d)).astype(np.float32)
    gi = rng.standard_normal((I, d)).astype(np.float32)
    return m, u0, i0, gu, gi


@pytest.mark.parametrize("d,L,tuning", [(64, 2, None), (32, 1, (4, 8, 64)), (128, 3, (8, 16, 64)), (64, 3, (2, 4, 64))])
def test_gnn_interval_backward_vs_autograd(dev, d, L, tuning):
    from sa_gnn_amd import graph, ops
    rng = np.random.default_rng(d * 10 + L)
    U, I = 157, 211
    m, u0, i0, gu, gi = _case(rng, U, I, 0.07, d)
    m = m.tolil()
    m[5, :] = 0                      # an isolated user: s = 0 exactly, the tie case of tf.maximum
    m = sp.csr_matrix(m)
    adj_idx, tp_idx = O.trans_to_lsts(m)[0], O.trans_to_lsts(O.transpose(m))[0]
    # oracle: torch autograd in float64
    tu = torch.tensor(u0, dtype=torch.float64, requires_grad=True)
    ti = torch.tensor(i0, dtype=torch.float64, requires_grad=True)
    ou, oi = O.torch_gnn_interval(tu, ti, adj_idx, tp_idx, L, 0.5)
    (ou * torch.tensor(gu, dtype=torch.float64)).sum().add((oi * torch.tensor(gi, dtype=torch.float64)).sum()).backward()
    # HIP path
    fwd, tp = graph.interval_pair(m, dev, tuning=tuning)
    mask_u = torch.empty((L, U, d // 4), dtype=torch.uint8, device=dev)
    mask_i = torch.empty((L, I, d // 4), dtype=torch.uint8, device=dev)
    uo = torch.empty((U, d), device=dev)
    io = torch.empty((I, d), device=dev)
    ops.gnn_interval(fwd.plan, tp.plan, torch.from_numpy(u0).to(dev), torch.from_numpy(i0).to(dev), L, 0.5, uo, io,
                     mask_u=mask_u, mask_i=mask_i)
    np.testing.assert_allclose(uo.cpu().numpy(), ou.detach().numpy(), rtol=1e-4, atol=1e-4)
    du, di = ops.gnn_interval_bwd(fwd.plan, tp.plan, torch.from_numpy(gu).to(dev), torch.from_numpy(gi).to(dev),
                                  L, 0.5, mask_u, mask_i)
    scale = max(float(tu.grad.abs().max()), 1.0)
    np.testing.assert_allclose(du.cpu().numpy(), tu.grad.numpy(), rtol=1e-4, atol=2e-6 * scale * L * 50)
    np.testing.assert_allclose(di.cpu().numpy(), ti.grad.numpy(), rtol=1e-4, atol=2e-6 * scale * L * 50)


def test_autograd_function_end_to_end(dev):
    """GnnIntervalFn inside a torch graph: gradients of a scalar loss w.r.t. the embedding tables."""
    from sa_gnn_amd import autograd as ag
    from sa_gnn_amd import graph
    rng = np.random.default_rng(3)
    U, I, d, L = 90, 120, 64, 2
    m, u0, i0, _, _ = _case(rng, U, I, 0.1, d)
    fwd, tp = graph.interval_pair(m, dev)
    pu = torch.from_numpy(u0).to(dev).requires_grad_(True)
    pi = torch.from_numpy(i0).to(dev).requires_grad_(True)
    uo, io = ag.gnn_interval(pu, pi, fwd.plan, tp.plan, L, 0.5)
    loss = (uo ** 2).sum() * 0.5 + (io[:, :8] * 3.0).sum()
    loss.backward()
    tu = torch.tensor(u0, dtype=torch.float64, requires_grad=True)
    ti = torch.tensor(i0, dtype=torch.float64, requires_grad=True)
    ou, oi = O.torch_gnn_interval(tu, ti, O.trans_to_lsts(m)[0], O.trans_to_lsts(O.transpose(m))[0], L, 0.5)
    ((ou ** 2).sum() * 0.5 + (oi[:, :8] * 3.0).sum()).backward()
    s = float(tu.grad.abs().max())
    np.testing.assert_allclose(pu.grad.cpu().numpy(), tu.grad.numpy(), rtol=2e-4, atol=1e-5 * s)
    np.testing.assert_allclose(pi.grad.cpu().numpy(), ti.grad.numpy(), rtol=2e-4, atol=1e-5 * s)


@pytest.mark.parametrize("d,t,n,heads", [(64, 3, 300, 16), (32, 2, 130, 16), (64, 1, 70, 16), (64, 5, 97, 4), (128, 3, 90, 16),
                                          (64, 8, 41, 16), (32, 6, 53, 16), (64, 12, 19, 16), (64, 16, 23, 16), (32, 12, 31, 16), (32, 16, 17, 16),
                                          (64, 2, 1000, 16), (64, 4, 77, 16), (64, 5, 61, 16), (64, 6, 37, 16),
                                          (32, 1, 33, 16), (32, 3, 90, 16), (32, 4, 70, 16), (32, 5, 45, 16), (32, 8, 29, 16)])
def test_interval_fusion_backward_vs_autograd(dev, d, t, n, heads):
    """Every gradient of the fusion (x and all ten parameter tensors) against float64 autograd."""
    from sa_gnn_amd import autograd as ag
    rng = np.random.default_rng(d + t + n)
    x = rng.standard_normal((n, t, d)).astype(np.float32)
    p = O.init_fusion_params(d, rng)
    gout = rng.standard_normal((n, d)).astype(np.float32)
    # oracle
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    out = O.torch_interval_fusion(tx, tp, heads)
    (out * torch.tensor(gout, dtype=torch.float64)).sum().backward()
    # HIP path; x given as a [t, n, d] storage viewed [n, t, d] (the exchange layout)
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(1, 0, 2))).to(dev).permute(1, 0, 2).requires_grad_(True)
    pd = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in p.items()}
    got = ag.interval_fusion(xd, pd, heads)
    np.testing.assert_allclose(got.detach().cpu().numpy(), out.detach().numpy(), rtol=1e-4, atol=2e-5)
    got.backward(torch.from_numpy(gout).to(dev))

    def check(name, a, b):
        a, b = a.detach().cpu().numpy().astype(np.float64), b.detach().numpy()
        # relative bar 1e-4 plus an absolute floor: some gradients are analytically ~0 (a key bias
        # shifts every score of a row alike; only the 1e-8 in the normaliser breaks the symmetry),
        # what remains of them is fp32 accumulation noise over n*t rows: eps32 * sqrt(n*t), 8e-6 at n*t ~ 1e3 (the rule of
        # test_training_forward_and_backward_many_tiles_per_block; a random sweep of shapes — tools/fuzz_gpu.py — meets
        # |dWk| ~ 9e-6 with an error of 5.02e-6 at n*t ~ 2e3)
        tol = 1e-4 * np.abs(b) + max(2e-5 * np.abs(b).max(), 8e-6 * max(1.0, np.sqrt(n * t / 1000.0)))
        bad = np.abs(a - b) > tol
        assert not bad.any(), f"{name}: {bad.sum()}/{bad.size} off, worst {np.abs(a - b)[bad].max():.3e} (scale {np.abs(b).max():.3e})"

    check("dx", xd.grad, tx.grad)
    for k in p:
        check("d" + k, pd[k].grad, tp[k].grad)


def test_interval_fusion_backward_with_output_dropout(dev):
    """DropoutWrapper(output_keep_prob): the mask scales the emitted h only; gradients follow."""
    from sa_gnn_amd import autograd as ag
    rng = np.random.default_rng(77)
    n, t, d, heads = 150, 3, 64, 16
    x = rng.standard_normal((n, t, d)).astype(np.float32)
    p = O.init_fusion_params(d, rng)
    scale = ((rng.random((n, t, d)) < 0.5) * 2.0).astype(np.float32)
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    h = O.torch_basic_lstm(tx, tp["lstm_W"], tp["lstm_b"]) * torch.tensor(scale, dtype=torch.float64)
    out = O.torch_mhsa_mean(O.torch_layer_norm_td(h, tp["ln_gamma"], tp["ln_beta"]), tp["Wq"], tp["bq"], tp["Wk"],
                            tp["bk"], tp["Wv"], tp["bv"], heads)
    out.square().sum().backward()
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    pd = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in p.items()}
    got = ag.interval_fusion(xd, pd, heads, drop_scale=torch.from_numpy(scale).to(dev))
    np.testing.assert_allclose(got.detach().cpu().numpy(), out.detach().numpy(), rtol=1e-4, atol=2e-5)
    got.square().sum().backward()
    for name, a, b in [("dx", xd.grad, tx.grad)] + [("d" + k, pd[k].grad, tp[k].grad) for k in p]:
        a, b = a.cpu().numpy().astype(np.float64), b.numpy()
        tol = 1e-4 * np.abs(b) + max(2e-5 * np.abs(b).max(), 2e-5)   # floor: see the test above
        assert (np.abs(a - b) <= tol).all(), name


@pytest.mark.parametrize("d,t,n", [(64, 4, 20011), (32, 3, 33000)])
def test_fused_bptt_matches_step_loop(dev, d, t, n, monkeypatch):
    """sagnn_lstm_bwd_f32 (one launch, gate gradients on chip) against the per-step entries it
    replaces, at sizes where every block walks several chunks and the last chunk is ragged."""
    from sa_gnn_amd import autograd as ag
    rng = np.random.default_rng(d * t)
    p = O.init_fusion_params(d, rng)
    x = torch.from_numpy(rng.standard_normal((n, t, d)).astype(np.float32)).to(dev)
    gout = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(dev)
    scale = torch.from_numpy(((rng.random((n, t, d)) < 0.7) / 0.7).astype(np.float32)).to(dev)
    grads = {}
    for mode in ("fused", "steps"):
        monkeypatch.setattr(ag, "FUSED_BPTT", mode == "fused")
        xd = x.clone().requires_grad_(True)
        pd = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in p.items()}
        ag.interval_fusion(xd, pd, 16, drop_scale=scale).backward(gout)
        grads[mode] = {"x": xd.grad, **{k: pd[k].grad for k in ("lstm_W", "lstm_b")}}
    for k in grads["fused"]:
        a, b = grads["fused"][k].double().cpu().numpy(), grads["steps"][k].double().cpu().numpy()
        tol = 1e-4 * np.abs(b) + 2e-5 * np.abs(b).max()
        assert (np.abs(a - b) <= tol).all(), f"{k}: worst {np.abs(a - b).max():.3e} (scale {np.abs(b).max():.3e})"


def test_adam_step_matches_tf_formula(dev):
    from sa_gnn_amd import ops
    rng = np.random.default_rng(1)
    p0 = rng.standard_normal(1000).astype(np.float32)
    params = {"w": torch.from_numpy(p0.copy()).to(dev)}
    opt = ops.Adam(params, lr=1e-2, decay=0.96, decay_step=2, reg=1e-2, reg_names={"w"})
    p, m, v = p0.astype(np.float64), np.zeros(1000), np.zeros(1000)
    for step in range(1, 6):
        g = rng.standard_normal(1000).astype(np.float32)
        opt.step({"w": torch.from_numpy(g).to(dev)})
        lr = 1e-2 * 0.96 ** ((step - 1) // 2)
        gg = g + 2 * 1e-2 * p
        m = 0.9 * m + 0.1 * gg
        v = 0.999 * v + 0.001 * gg * gg
        p = p - lr * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step) * m / (np.sqrt(v) + 1e-8)
    np.testing.assert_allclose(params["w"].cpu().numpy(), p, rtol=1e-4, atol=1e-5)


def test_adam_multi_more_tensors_than_one_table_with_an_empty_one(dev):
    """sagnn_adam_multi_f32 walks its tensors in tables of 48 non-empty ones. 60 tensors with an empty one in the
    first table: the second launch must start where the first stopped — every tensor takes exactly ONE step."""
    from sa_gnn_amd import ops
    rng = np.random.default_rng(5)
    sizes = [int(s) for s in rng.integers(4, 3000, size=60)]
    sizes[10] = 0
    p0 = [rng.standard_normal(s).astype(np.float32) for s in sizes]
    g0 = [rng.standard_normal(s).astype(np.float32) for s in sizes]
    params = {f"w{i}": torch.from_numpy(a.copy()).to(dev) for i, a in enumerate(p0)}
    opt = ops.Adam(params, lr=1e-2, reg=1e-2, reg_names={f"w{i}" for i in range(0, 60, 2)})
    opt.step({f"w{i}": torch.from_numpy(g).to(dev) for i, g in enumerate(g0)})
    for i in range(60):
        gg = g0[i].astype(np.float64) + (2 * 1e-2 * p0[i] if i % 2 == 0 else 0.0)
        m, v = 0.1 * gg, 0.001 * gg * gg
        want = p0[i] - 1e-2 * np.sqrt(1 - 0.999) / (1 - 0.9) * m / (np.sqrt(v) + 1e-8)
        np.testing.assert_allclose(params[f"w{i}"].cpu().numpy(), want, rtol=1e-4, atol=1e-5, err_msg=f"tensor {i}")


def test_gnn_interval_backward_with_duplicated_stored_entries(dev):
    """A subMat with a duplicated stored (u, i): the forward counts it twice on the user side and once
    on the item side (DataHandler.transpose merges it, DataHandler.py:9-11), so the two patterns are
    not transposes of each other and the backward needs the exact adjoints (graph.interval_pair)."""
    from sa_gnn_amd import graph, ops
    rng = np.random.default_rng(99)
    U, I, d, L = 61, 83, 64, 2
    base = sp.csr_matrix((rng.random((U, I)) < 0.08).astype(np.intc))
    indptr, indices = base.indptr.copy(), base.indices.copy()
    # duplicate the first stored entry of rows 3, 10 and 40 (and one of them twice)
    rows, cols = [], []
    for r in range(U):
        cs = list(indices[indptr[r]:indptr[r + 1]])
        if r in (3, 10, 40) and cs:
            cs = [cs[0]] * (3 if r == 10 else 2) + cs[1:]
        rows += [r] * len(cs)
        cols += cs
    ptr = np.zeros(U + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=U), out=ptr[1:])
    m = sp.csr_matrix((np.ones(len(cols), dtype=np.intc), np.asarray(cols, dtype=np.int32), ptr), shape=(U, I))
    adj_idx, tp_idx = O.trans_to_lsts(m)[0], O.trans_to_lsts(O.transpose(m))[0]
    assert len(adj_idx) > len(tp_idx)                        # the quirk is present
    u0 = rng.standard_normal((U, d)).astype(np.float32)
    i0 = rng.standard_normal((I, d)).astype(np.float32)
    gu = rng.standard_normal((U, d)).astype(np.float32)
    gi = rng.standard_normal((I, d)).astype(np.float32)
    tu = torch.tensor(u0, dtype=torch.float64, requires_grad=True)
    ti = torch.tensor(i0, dtype=torch.float64, requires_grad=True)
    ou, oi = O.torch_gnn_interval(tu, ti, adj_idx, tp_idx, L, 0.5)
    ((ou * torch.tensor(gu, dtype=torch.float64)).sum() + (oi * torch.tensor(gi, dtype=torch.float64)).sum()).backward()
    fwd, tp = graph.interval_pair(m, dev)
    assert fwd.plan.partner_adjoint is not None and tp.plan.partner_adjoint is not None
    mask_u = torch.empty((L, U, d // 4), dtype=torch.uint8, device=dev)
    mask_i = torch.empty((L, I, d // 4), dtype=torch.uint8, device=dev)
    uo, io = torch.empty((U, d), device=dev), torch.empty((I, d), device=dev)
    ops.gnn_interval(fwd.plan, tp.plan, torch.from_numpy(u0).to(dev), torch.from_numpy(i0).to(dev), L, 0.5, uo, io,
                     mask_u=mask_u, mask_i=mask_i)
    np.testing.assert_allclose(uo.cpu().numpy(), ou.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(io.cpu().numpy(), oi.detach().numpy(), rtol=1e-4, atol=1e-4)
    du, di = ops.gnn_interval_bwd(fwd.plan, tp.plan, torch.from_numpy(gu).to(dev), torch.from_numpy(gi).to(dev),
                                  L, 0.5, mask_u, mask_i)
    scale = max(float(tu.grad.abs().max()), 1.0)
    np.testing.assert_allclose(du.cpu().numpy(), tu.grad.numpy(), rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(di.cpu().numpy(), ti.grad.numpy(), rtol=1e-4, atol=1e-4 * scale)
    # a hand-made pair without the adjoints is refused instead of giving silently wrong gradients
    fwd.plan.partner_adjoint = tp.plan.partner_adjoint = None
    with pytest.raises(ValueError, match="transposed pair"):
        ops.gnn_interval_bwd(fwd.plan, tp.plan, torch.from_numpy(gu).to(dev), torch.from_numpy(gi).to(dev), L, 0.5,
                             mask_u, mask_i)
