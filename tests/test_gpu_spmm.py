"""GPU parity: the HIP interval SpMM (through the C ABI) against the oracle on the same seeded
inputs. fp32; tolerance 1e-4 relative / 1e-5 absolute (north_star: within 1e-4 fp32) — the only
difference allowed is summation order."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import GOLDEN
from oracle import selfgnn_oracle as O

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 1e-5


def assert_sum_close(got, want, abs_terms):
    """|got - want| <= 1e-4*|want| + 1e-5 + 3*eps32*sum|terms|: the relative bar of north_star plus
    the fp32 forward-error allowance for an element that cancels to ~0 out of large terms (the
    kernel and the oracle add the same terms in different orders)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    tol = RTOL * np.abs(want) + ATOL + 2e-7 * np.asarray(abs_terms, np.float64)
    bad = np.abs(got - want) > tol
    assert not bad.any(), (f"{bad.sum()} / {bad.size} elements outside tolerance; worst "
                           f"{np.abs(got - want)[bad].max():.3e} vs tol {tol[bad].min():.3e}")


def _graph(rng, n_rows, n_src, degs):
    cols = [np.sort(rng.choice(n_src, size=int(dg), replace=False)) if dg <= n_src else
            rng.integers(0, n_src, size=int(dg)) for dg in degs]
    rowptr = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int32)
    colidx = (np.concatenate(cols) if len(cols) and rowptr[-1] else np.zeros(0)).astype(np.int32)
    idx = np.stack([np.repeat(np.arange(n_rows), np.diff(rowptr)), colidx], 1).astype(np.int32)
    return rowptr, colidx, idx


def _oracle(idx, x, n_rows, leaky, residual=None, acc_in=None):
    y = O.message_propagate_zero_fill(x, idx, n_rows, leaky) if len(idx) else np.zeros((n_rows, x.shape[1]), np.float32)
    if residual is not None:
        y = y + residual
    acc = y if acc_in is None else acc_in + y
    return y, acc


@pytest.mark.parametrize("d", [32, 64, 128, 256, 48, 4])
def test_spmm_degree_classes(dev, d):
    """Short, medium and long (chunked + fix-up) rows in one matrix, thresholds lowered so every
    class appears at a size the oracle finishes instantly."""
    from sa_gnn_amd import ops
    rng = np.random.default_rng(100 + d)
    n_rows, n_src = 777, 1500
    degs = rng.integers(0, 12, size=n_rows)
    degs[rng.choice(n_rows, 60, replace=False)] = rng.integers(13, 200, size=60)
    degs[[5, 400, 776]] = [1400, 900, 1500]
    degs[[0, 1, 2, 100, 775]] = 0
    rowptr, colidx, idx = _graph(rng, n_rows, n_src, degs)
    x = rng.standard_normal((n_src, d)).astype(np.float32)
    res = rng.standard_normal((n_rows, d)).astype(np.float32)
    acc0 = rng.standard_normal((n_rows, d)).astype(np.float32)
    xd, rd, ad = (torch.from_numpy(a).to(dev) for a in (x, res, acc0))
    for tuning in (None, (8, 128, 64), (16, 256, 128), (2, 4, 64)):
        plan = ops.SpmmPlan(rowptr, colidx, n_rows, n_src, device=dev, tuning=tuning)
        if tuning is not None:
            assert plan.info.n_long_rows > 0
        want_y, want_acc = _oracle(idx, x, n_rows, 0.5, res, acc0)
        terms, _ = _oracle(idx, np.abs(x), n_rows, 1.0)
        acc = torch.empty_like(rd)
        y = ops.spmm(plan, xd, 0.5, residual=rd, acc_in=ad, acc_out=acc)
        assert_sum_close(y.cpu().numpy(), want_y, terms)
        assert_sum_close(acc.cpu().numpy(), want_acc, terms)
        # plain form: no residual, no accumulator; isolated rows are exactly zero
        y2 = ops.spmm(plan, xd, 0.1)
        w2, _ = _oracle(idx, x, n_rows, 0.1)
        assert_sum_close(y2.cpu().numpy(), w2, terms)
        assert torch.all(y2[[0, 1, 2, 100, 775]] == 0)


def test_spmm_short_rows_bit_exact(dev):
    """Short rows use one accumulator in edge order = TF's sequential SegmentSum order."""
    from sa_gnn_amd import ops
    rng = np.random.default_rng(9)
    n_rows, n_src, d = 300, 64, 64
    rowptr, colidx, idx = _graph(rng, n_rows, n_src, rng.integers(0, 17, size=n_rows))
    x = rng.standard_normal((n_src, d)).astype(np.float32)
    plan = ops.SpmmPlan(rowptr, colidx, n_rows, n_src, device=dev)
    y = ops.spmm(plan, torch.from_numpy(x).to(dev), 0.5)
    want = O.leaky_relu(np.concatenate([O.segment_sum(x[idx[:, 1]], idx[:, 0]),
                                        np.zeros((n_rows, d), np.float32)])[:n_rows], 0.5)
    np.testing.assert_array_equal(y.cpu().numpy(), want)


def test_spmm_edge_cases(dev):
    from sa_gnn_amd import graph, ops
    rng = np.random.default_rng(2)
    d = 64
    # empty interval -> phantom (0,0) edge: out[0] = leaky(x[0])
    adj = graph.IntervalAdj.from_scipy(sp.csr_matrix((9, 7), dtype=np.intc), dev)
    x = torch.randn(7, d, device=dev)
    y = ops.spmm(adj.plan, x, 0.5)
    torch.testing.assert_close(y[0], torch.maximum(0.5 * x[0], x[0]))
    assert torch.all(y[1:] == 0)
    # >100 trailing empty rows: exactly [N, d], zeros (the reference's TF-GPU behaviour)
    m = sp.csr_matrix(([1, 1], ([0, 3], [2, 5])), shape=(400, 7), dtype=np.intc)
    adj = graph.IntervalAdj.from_scipy(m, dev)
    y = ops.spmm(adj.plan, x, 0.5)
    assert y.shape == (400, d) and torch.all(y[4:] == 0)
    # duplicated stored entry counts twice forward, once transposed (scipy merges it)
    m = sp.csr_matrix((np.array([1, 1, 1], np.intc), np.array([2, 2, 4], np.int32), np.array([0, 3, 3], np.int32)), shape=(2, 7))
    fwd, tp = graph.interval_pair(m, dev)
    y = ops.spmm(fwd.plan, x, 1.0)
    torch.testing.assert_close(y[0], 2 * x[2] + x[4])
    xu = torch.randn(2, d, device=dev)
    yt = ops.spmm(tp.plan, xu, 1.0)
    torch.testing.assert_close(yt[2], xu[0])
    # n_rows not a multiple of the row block; single row; strided views
    for n_rows in (1, 15, 16, 17, 63, 65):
        rowptr, colidx, idx = _graph(rng, n_rows, 7, rng.integers(0, 7, size=n_rows))
        plan = ops.SpmmPlan(rowptr, colidx, n_rows, 7, device=dev)
        slab = torch.full((n_rows, 3, d), 7.0, device=dev)
        ops.spmm(plan, x, 0.5, out=slab[:, 1, :])
        want, _ = _oracle(idx, x.cpu().numpy(), n_rows, 0.5)
        np.testing.assert_allclose(slab[:, 1, :].cpu().numpy(), want, rtol=RTOL, atol=ATOL)
        assert torch.all(slab[:, 0, :] == 7) and torch.all(slab[:, 2, :] == 7)
    # argument errors surface as exceptions, not wrong answers
    from sa_gnn_amd._lib import SagnnError
    with pytest.raises(ValueError):
        ops.spmm(plan, torch.randn(8, d, device=dev), 0.5)
    with pytest.raises(SagnnError):
        ops.spmm(plan, x[:, :62].contiguous(), 0.5)
    with pytest.raises(SagnnError):
        ops.spmm(plan, x, 0.5, want_out=False)


@pytest.mark.parametrize("d,L", [(64, 2), (32, 1), (128, 3)])
def test_gnn_interval_vs_oracle(dev, d, L):
    from sa_gnn_amd import graph, ops
    rng = np.random.default_rng(40 + d)
    U, I, T = 211, 157, 3
    m = sp.csr_matrix((rng.random((U, I)) < 0.06).astype(np.intc))
    fwd, tp = graph.interval_pair(m, dev, tuning=(8, 16, 64))
    u0 = rng.standard_normal((U, d)).astype(np.float32)
    i0 = rng.standard_normal((I, d)).astype(np.float32)
    adj_idx, tp_idx = O.trans_to_lsts(m)[0], O.trans_to_lsts(O.transpose(m))[0]
    want_u, want_i = O.gnn_interval(u0, i0, adj_idx, tp_idx, L, 0.5)
    terms_u, terms_i = O.gnn_interval(np.abs(u0), np.abs(i0), adj_idx, tp_idx, L, 1.0)
    us = torch.zeros((U, T, d), device=dev)
    its = torch.zeros((I, T, d), device=dev)
    ops.gnn_interval(fwd.plan, tp.plan, torch.from_numpy(u0).to(dev), torch.from_numpy(i0).to(dev), L, 0.5,
                     us[:, 1, :], its[:, 1, :])
    assert_sum_close(us[:, 1].cpu().numpy(), want_u, terms_u)
    assert_sum_close(its[:, 1].cpu().numpy(), want_i, terms_i)
    assert torch.all(us[:, 0] == 0) and torch.all(us[:, 2] == 0)


@pytest.mark.parametrize("d", [32, 64, 128])
def test_gnn_stack_golden(dev, d):
    """Committed vectors (tests/golden/oracle_tiny.npz): T=3, L=2, leaky 0.5."""
    from sa_gnn_amd import graph, ops
    g = np.load(os.path.join(GOLDEN, "oracle_tiny.npz"))
    U, I, T = 37, 53, 3
    us = torch.empty((U, T, d), device=dev)
    its = torch.empty((I, T, d), device=dev)
    for k in range(T):
        adj, tpi = g[f"d{d}/adj{k}"], g[f"d{d}/tp{k}"]
        rp = np.concatenate([[0], np.cumsum(np.bincount(adj[:, 0], minlength=U))]).astype(np.int32)
        rpt = np.concatenate([[0], np.cumsum(np.bincount(tpi[:, 0], minlength=I))]).astype(np.int32)
        pu = ops.SpmmPlan(rp, adj[:, 1].copy(), U, I, device=dev)
        pi = ops.SpmmPlan(rpt, tpi[:, 1].copy(), I, U, device=dev)
        ops.gnn_interval(pu, pi, torch.from_numpy(g[f"d{d}/uEmbed"][k]).to(dev),
                         torch.from_numpy(g[f"d{d}/iEmbed"][k]).to(dev), 2, 0.5, us[:, k, :], its[:, k, :])
    np.testing.assert_allclose(us.cpu().numpy(), g[f"d{d}/user_vector"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(its.cpu().numpy(), g[f"d{d}/item_vector"], rtol=RTOL, atol=ATOL)


def test_spmm_powerlaw_properties_large(dev):
    """Size-independent properties at a size the oracle would not finish quickly: linearity in X
    (leaky = 1 makes the op linear), agreement between the transposed pair through
    <A x, y> = <x, A^T y>, and row sums against degrees."""
    from sa_gnn_amd import ops, synthetic
    U, I, d = 200_000, 100_000, 64
    u, i = synthetic.powerlaw_edges(U, I, 2_000_000, seed=1000, device=dev)
    (rp_u, ci_u), (rp_i, ci_i) = synthetic.csr_pair_from_edges(u, i, U, I)
    pu = ops.SpmmPlan(rp_u, ci_u, U, I, device=dev, validate=False)
    pi = ops.SpmmPlan(rp_i, ci_i, I, U, device=dev, validate=False)
    assert pi.info.n_long_rows > 0                       # hub items exercise the chunked path
    ones = torch.ones((I, d), device=dev)
    deg = (rp_u[1:] - rp_u[:-1]).to(torch.float32)
    torch.testing.assert_close(ops.spmm(pu, ones, 1.0)[:, 0], deg)
    torch.testing.assert_close(ops.spmm(pi, torch.ones((U, d), device=dev), 1.0)[:, 5], (rp_i[1:] - rp_i[:-1]).to(torch.float32))
    x1, x2 = torch.randn(I, d, device=dev), torch.randn(I, d, device=dev)
    lhs = ops.spmm(pu, 2 * x1 - 3 * x2, 1.0)
    rhs = 2 * ops.spmm(pu, x1, 1.0) - 3 * ops.spmm(pu, x2, 1.0)
    torch.testing.assert_close(lhs, rhs, rtol=1e-4, atol=1e-3)
    y = torch.randn(U, d, device=dev)
    a = (ops.spmm(pu, x1, 1.0).double() * y.double()).sum()
    b = (x1.double() * ops.spmm(pi, y, 1.0).double()).sum()
    assert abs(a - b) <= 1e-6 * max(abs(a), abs(b), 1.0) + 1e-2
    # leaky applies to the sum, not per edge: negative sums scale by the slope
    s = ops.spmm(pu, x1, 1.0)
    torch.testing.assert_close(ops.spmm(pu, x1, 0.5), torch.maximum(0.5 * s, s))


def _intervals(rng, U, I, T, dens):
    mats = []
    for k in range(T):
        m = (rng.random((U, I)) < dens[k % len(dens)]).astype(np.intc)
        m[3, :] = 1                       # a user with every item: a long row of the user side
        m[:, 5] = 1                       # an item with every user: a long row of the item side
        if k == 1:
            m[10:40, :] = 0               # isolated users in this interval only
        mats.append(sp.csr_matrix(m))
    return mats


@pytest.mark.parametrize("d,L,T", [(64, 2, 3), (32, 1, 5), (128, 3, 4), (64, 3, 5)])
def test_gnn_stack_one_launch_per_layer_vs_oracle(dev, d, L, T):
    """sagnn_gnn_stack_f32: every interval of the loop of model.py:118-129 in ONE launch per layer (+ one fix-up
    launch), all three degree classes in every interval (tuning 8 / 24 / 64), outputs written as columns of the
    [N, T, d] tensors the fusion reads. Against O.gnn_stack, and bit for bit against the per-interval entry
    (same kernel bodies, same order of additions)."""
    from sa_gnn_amd import graph, ops
    rng = np.random.default_rng(7 * d + L + T)
    U, I = 301, 211
    mats = _intervals(rng, U, I, T, (0.04, 0.09, 0.02))
    pairs = [graph.interval_pair(m, dev, tuning=(8, 24, 64)) for m in mats]
    ue = rng.standard_normal((T, U, d)).astype(np.float32)
    ie = rng.standard_normal((T, I, d)).astype(np.float32)
    batch = ops.SpmmBatch([p[0].plan for p in pairs], [p[1].plan for p in pairs])
    assert all(p[0].plan.info.n_long_rows > 0 and p[1].plan.info.n_long_rows > 0 for p in pairs)
    ued, ied = torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev)
    us = torch.full((U, T, d), 7.0, device=dev)
    its = torch.full((I, T, d), 7.0, device=dev)
    lib = ops._lib.load()
    lib.sagnn_profile_enable(64)
    ops.gnn_stack(batch, ued, ied, L, 0.5, us.permute(1, 0, 2), its.permute(1, 0, 2))
    import ctypes
    n, kinds = ctypes.c_int(0), (ctypes.c_int32 * 64)()
    ops.check(lib.sagnn_profile_read(None, kinds, None, None, 64, ctypes.byref(n)))
    lib.sagnn_profile_enable(0)
    assert [kinds[i] for i in range(n.value)] == [0, 1] * L              # one row launch + one fix-up per layer
    adjs = [O.trans_to_lsts(m)[0] for m in mats]
    tps = [O.trans_to_lsts(O.transpose(m))[0] for m in mats]
    want_u, want_i = O.gnn_stack(ue, ie, adjs, tps, L, 0.5)              # [U, T, d], [I, T, d]
    terms_u, terms_i = O.gnn_stack(np.abs(ue), np.abs(ie), adjs, tps, L, 1.0)
    assert_sum_close(us.cpu().numpy(), want_u, terms_u)
    assert_sum_close(its.cpu().numpy(), want_i, terms_i)
    us2 = torch.empty((U, T, d), device=dev)
    its2 = torch.empty((I, T, d), device=dev)
    for k in range(T):
        ops.gnn_interval(pairs[k][0].plan, pairs[k][1].plan, ued[k], ied[k], L, 0.5, us2[:, k, :], its2[:, k, :])
    assert torch.equal(us, us2) and torch.equal(its, its2)
    # [T, N, d] storage as well (the training path's slabs), and the recorded masks
    out_u, out_i = torch.empty((T, U, d), device=dev), torch.empty((T, I, d), device=dev)
    mu = torch.zeros((T, L, U, d // 4), dtype=torch.uint8, device=dev)
    mi = torch.zeros((T, L, I, d // 4), dtype=torch.uint8, device=dev)
    ops.gnn_stack(batch, ued, ied, L, 0.5, out_u, out_i, mask_u=mu, mask_i=mi)
    assert torch.equal(out_u.permute(1, 0, 2), us) and torch.equal(out_i.permute(1, 0, 2), its)
    mu2 = torch.zeros((L, U, d // 4), dtype=torch.uint8, device=dev)
    mi2 = torch.zeros((L, I, d // 4), dtype=torch.uint8, device=dev)
    for k in range(T):
        ops.gnn_interval(pairs[k][0].plan, pairs[k][1].plan, ued[k], ied[k], L, 0.5, us2[:, k, :], its2[:, k, :], mask_u=mu2, mask_i=mi2)
        assert torch.equal(mu[k], mu2) and torch.equal(mi[k], mi2)


def test_gnn_stack_batch_with_an_empty_interval(dev):
    """The batched entry over intervals of which one stores NO entry (the reference's transToLsts then holds the phantom
    (0, 0) edge: SURVEY §0.4) and one has only trailing-empty rows: bit for bit the per-interval entry, forward and
    backward (the adjoint of the phantom edge reaches row 0 of the other node type)."""
    from sa_gnn_amd import autograd as ag
    from sa_gnn_amd import graph, ops
    rng = np.random.default_rng(5)
    U, I, T, L, d = 137, 95, 3, 2, 64
    mats = _intervals(rng, U, I, T, (0.05,))
    mats[1] = sp.csr_matrix((U, I), dtype=np.intc)                          # nothing happened in interval 1
    mats[2] = sp.csr_matrix((np.ones(2, np.intc), (np.array([0, 3]), np.array([2, 5]))), shape=(U, I))
    pairs = [graph.interval_pair(m, dev, tuning=(8, 24, 64)) for m in mats]
    batch = ops.SpmmBatch([p[0].plan for p in pairs], [p[1].plan for p in pairs])
    ue = torch.from_numpy(rng.standard_normal((T, U, d)).astype(np.float32)).to(dev)
    ie = torch.from_numpy(rng.standard_normal((T, I, d)).astype(np.float32)).to(dev)
    gu = torch.from_numpy(rng.standard_normal((T, U, d)).astype(np.float32)).to(dev)
    gi = torch.from_numpy(rng.standard_normal((T, I, d)).astype(np.float32)).to(dev)
    res = []
    for plans in ((batch, None), ([p[0].plan for p in pairs], [p[1].plan for p in pairs])):
        a, b = ue.clone().requires_grad_(True), ie.clone().requires_grad_(True)
        ou, oi = ag.gnn_stack(a, b, plans[0], plans[1], L, 0.5)
        ((ou * gu).sum() + (oi * gi).sum()).backward()
        res.append((ou.detach(), oi.detach(), a.grad, b.grad))
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)
    ou = res[0][0]
    adjs = [O.trans_to_lsts(m)[0] for m in mats]
    tps = [O.trans_to_lsts(O.transpose(m))[0] for m in mats]
    want_u, want_i = O.gnn_stack(ue.cpu().numpy(), ie.cpu().numpy(), adjs, tps, L, 0.5)     # [U, T, d]
    np.testing.assert_allclose(ou.permute(1, 0, 2).cpu().numpy(), want_u, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(res[0][1].permute(1, 0, 2).cpu().numpy(), want_i, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("d,L,T", [(64, 2, 3), (32, 3, 4)])
def test_gnn_stack_backward_matches_per_interval_backward(dev, d, L, T):
    """sagnn_gnn_stack_bwd_f32 against sagnn_gnn_interval_bwd_f32 interval by interval (itself checked against
    float64 autograd in tests/test_gpu_backward.py), through the autograd node with a duplicated stored entry in one
    interval (exact adjoint plans)."""
    from sa_gnn_amd import autograd as ag
    from sa_gnn_amd import graph, ops
    rng = np.random.default_rng(11 * d + L)
    U, I = 157, 263
    mats = _intervals(rng, U, I, T, (0.05, 0.03))
    m = mats[0]
    indptr, indices = m.indptr.copy(), m.indices.copy()
    indices[indptr[7] + 1] = indices[indptr[7]]                          # row 7 stores one (u, i) twice
    mats[0] = sp.csr_matrix((np.ones(len(indices), np.intc), indices, indptr), shape=(U, I))
    pairs = [graph.interval_pair(mm, dev, tuning=(8, 24, 64)) for mm in mats]
    assert pairs[0][0].plan.partner_adjoint is not None
    batch = ops.SpmmBatch([p[0].plan for p in pairs], [p[1].plan for p in pairs])
    ue = torch.from_numpy(rng.standard_normal((T, U, d)).astype(np.float32)).to(dev)
    ie = torch.from_numpy(rng.standard_normal((T, I, d)).astype(np.float32)).to(dev)
    gu = torch.from_numpy(rng.standard_normal((U, T, d)).astype(np.float32)).to(dev)       # gradients arrive as [N, T, d] views
    gi = torch.from_numpy(rng.standard_normal((I, T, d)).astype(np.float32)).to(dev)
    res = []
    for plans in ((batch, None), ([p[0].plan for p in pairs], [p[1].plan for p in pairs])):
        a, b = ue.clone().requires_grad_(True), ie.clone().requires_grad_(True)
        ou, oi = ag.gnn_stack(a, b, plans[0], plans[1], L, 0.5)
        ((ou.permute(1, 0, 2) * gu).sum() + (oi.permute(1, 0, 2) * gi).sum()).backward()
        res.append((ou.detach(), oi.detach(), a.grad, b.grad))
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)
