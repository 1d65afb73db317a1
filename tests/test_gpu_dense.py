"""GPU parity of the dense MFMA building blocks against float64 matmul."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,din,dout", [(1000, 64, 192), (129, 192, 64), (5000, 256, 128), (31, 32, 96), (300, 96, 32), (4097, 128, 256), (700, 512, 128), (333, 128, 512), (200, 256, 384)])
def test_dense_nn(dev, n, din, dout):
    from sa_gnn_amd import ops
    g = torch.Generator(device="cpu").manual_seed(n + din)
    x = torch.randn((n, din), generator=g)
    W = torch.randn((din, dout), generator=g) / din ** 0.5
    b = torch.randn(dout, generator=g)
    want = x.double() @ W.double() + b.double()
    got = ops.dense_nn(x.to(dev), W.to(dev), b.to(dev))
    torch.testing.assert_close(got.cpu().double(), want, rtol=1e-4, atol=1e-5)
    # strided input/output views and accumulate
    big = torch.zeros((n, din + 64))
    big[:, 32:32 + din] = x
    out = torch.ones((n, dout + 32), device=dev)
    ops.dense_nn(big.to(dev)[:, 32:32 + din], W.to(dev), None, out=out[:, :dout], accumulate=True)
    torch.testing.assert_close(out[:, :dout].cpu().double(), x.double() @ W.double() + 1.0, rtol=1e-4, atol=1e-5)
    assert torch.all(out[:, dout:] == 1)


@pytest.mark.parametrize("n,din,dout", [(1000, 64, 192), (77, 128, 256), (20000, 128, 256), (513, 32, 96), (64, 192, 64), (999, 256, 128), (450, 256, 512), (300, 384, 128)])
def test_dense_tn(dev, n, din, dout):
    from sa_gnn_amd import ops
    gen = torch.Generator(device="cpu").manual_seed(n + dout)
    x = torch.randn((n, din), generator=gen)
    g = torch.randn((n, dout), generator=gen)
    dW = torch.zeros((din, dout), device=dev)
    db = torch.zeros(dout, device=dev)
    ops.dense_tn(x.to(dev), g.to(dev), dW, db)
    want = x.double().T @ g.double()
    scale = float(want.abs().max())
    torch.testing.assert_close(dW.cpu().double(), want, rtol=1e-4, atol=2e-6 * scale + 1e-5)
    torch.testing.assert_close(db.cpu().double(), g.double().sum(0), rtol=1e-4, atol=1e-4)
    ops.dense_tn(x.to(dev), g.to(dev), dW, None)                    # accumulates
    torch.testing.assert_close(dW.cpu().double(), 2 * want, rtol=1e-4, atol=4e-6 * scale + 1e-5)


@pytest.mark.parametrize("n,t,din,dout,layout", [(1000, 5, 128, 512, "node"), (333, 12, 128, 512, "time"), (77, 3, 64, 256, "node"),
                                                 (5000, 2, 32, 128, "time"), (19, 7, 96, 384, "node")])
def test_dense_tn_seg(dev, n, t, din, dout, layout):
    """sagnn_dense_tn_seg_f32: dW += sum_s x[s]^T g[s] in one launch, x node-major ([n, t, d] seen as [t, n, d]) or
    time-major, g time-major, and a SHIFTED pairing (x steps 0..t-2 against g steps 1..t-1: the h side of the BPTT)."""
    from sa_gnn_amd import ops
    gen = torch.Generator(device="cpu").manual_seed(n + t + dout)
    x = torch.randn((n, t, din), generator=gen)
    g = torch.randn((t, n, dout), generator=gen)
    xd = x.to(dev) if layout == "node" else x.to(dev).permute(1, 0, 2).contiguous().permute(1, 0, 2)
    assert xd.shape == (n, t, din)
    gd = g.to(dev)
    dW = torch.zeros((din, dout), device=dev)
    db = torch.zeros(dout, device=dev)
    ops.dense_tn_seg(xd.permute(1, 0, 2), gd, dW, db)
    want = torch.einsum("ntd,tne->de", x.double(), g.double())
    mag = torch.einsum("ntd,tne->de", x.double().abs(), g.double().abs())
    assert ((dW.cpu().double() - want).abs() <= 1e-4 * want.abs() + 1e-6 * mag + 1e-5).all()
    want_b = g.double().sum((0, 1))
    assert ((db.cpu().double() - want_b).abs() <= 1e-4 * want_b.abs() + 1e-6 * g.double().abs().sum((0, 1)) + 1e-5).all()
    if t > 1:
        dW2 = torch.zeros((din, dout), device=dev)
        ops.dense_tn_seg(xd.permute(1, 0, 2)[:t - 1], gd[1:], dW2, None)
        want2 = torch.einsum("ntd,tne->de", x[:, :t - 1].double(), g[1:].double())
        assert ((dW2.cpu().double() - want2).abs() <= 1e-4 * want2.abs() + 1e-6 * mag + 1e-5).all()
    with pytest.raises(ValueError):
        ops.dense_tn_seg(xd.permute(1, 0, 2), gd[:, : n - 1], dW, None)


@pytest.mark.parametrize("engine", ["f16x2", "f32"])
@pytest.mark.parametrize("rows,d", [(1000, 64), (31, 64), (40007, 64), (513, 32), (70001, 32)])
def test_attn_bwd_tail(dev, rows, d, engine):
    """sagnn_attn_bwd_tail_f32: dy (over y), dW and db of the three dense layers in one pass over
    dQ|dK|dV, against float64 matmuls; sizes from one ragged chunk to many chunks per block. Both engines:
    the f16 matrix cores (default) and the f32-MFMA kernel of round 1 (sagnn_set_engine(SAGNN_ENGINE_F32))."""
    from sa_gnn_amd import _lib, ops
    lib = _lib.load()
    gen = torch.Generator(device="cpu").manual_seed(rows + d)
    y = torch.randn((rows, d), generator=gen)
    dqkv = torch.randn((rows, 3 * d), generator=gen)
    W = torch.randn((d, 3 * d), generator=gen) / d ** 0.5
    yd, gd, Wd = y.to(dev), dqkv.to(dev), W.to(dev)
    dW = torch.zeros((d, 3 * d), device=dev)
    db = torch.zeros(3 * d, device=dev)
    assert lib.sagnn_attn_bwd_tail_supported(d)
    with ops.engine(engine):
        ops.check(lib.sagnn_attn_bwd_tail_f32(yd.data_ptr(), gd.data_ptr(), rows, d, Wd.data_ptr(), dW.data_ptr(),
                                              db.data_ptr(), None))
    want_dy = dqkv.double() @ W.double().T
    want_dW = y.double().T @ dqkv.double()
    torch.testing.assert_close(yd.cpu().double(), want_dy, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dW.cpu().double(), want_dW, rtol=1e-4, atol=2e-6 * float(want_dW.abs().max()) + 1e-5)
    # db sums `rows` terms of O(1) with float atomics (order varies run to run): fp32 rounding grows with the sum of the
    # terms' magnitudes — 2e-8 of it (a tenth of eps32 per term) next to 1e-4 of the value
    want_db = dqkv.double().sum(0)
    tol_b = 1e-4 * want_db.abs() + 2e-8 * dqkv.double().abs().sum(0) + 1e-5
    assert ((db.cpu().double() - want_db).abs() <= tol_b).all()
    assert torch.equal(gd.cpu(), dqkv)                                     # dQKV itself is read-only


@pytest.mark.parametrize("rows,d", [(1000, 64), (20011, 64), (777, 32)])
def test_attn_bwd_tail_beyond_the_f16_range(dev, rows, d):
    """The tail's products run on two-piece f16 operands. Gradient rows are scaled into the split's window by exact
    powers of two (a gradient of 1e6 or an isolated 2e30 is an ordinary row after that); a chunk of 32 rows that
    holds a y beyond the window (-3e5 here), or whose rows dwarf everything the block has met, takes no part in the
    matrix-core products: the block evaluates it with fp32 fmaf chains. dy, dW and db of those chunks, of their
    neighbours and of the rest must match."""
    from sa_gnn_amd import _lib, ops
    lib = _lib.load()
    gen = torch.Generator(device="cpu").manual_seed(3 * rows + d)
    y = torch.randn((rows, d), generator=gen)
    dqkv = torch.randn((rows, 3 * d), generator=gen)
    W = torch.randn((d, 3 * d), generator=gen) / d ** 0.5
    dqkv[5, 7] = 1e6
    y[rows // 2, 3] = -3e5
    dqkv[rows - 1, 3 * d - 1] = 2e30
    yd, gd, Wd = y.to(dev), dqkv.to(dev), W.to(dev)
    dW = torch.zeros((d, 3 * d), device=dev)
    db = torch.zeros(3 * d, device=dev)
    ops.check(lib.sagnn_attn_bwd_tail_f32(yd.data_ptr(), gd.data_ptr(), rows, d, Wd.data_ptr(), dW.data_ptr(),
                                          db.data_ptr(), None))
    want_dy = dqkv.double() @ W.double().T
    want_dW = y.double().T @ dqkv.double()
    want_db = dqkv.double().sum(0)
    assert torch.isfinite(yd).all() and torch.isfinite(dW).all()
    # the huge entries dominate their own rows / columns: relative to each output's own scale
    torch.testing.assert_close(yd.cpu().double(), want_dy, rtol=1e-4, atol=1e-5)
    tol_w = 1e-4 * want_dW.abs() + 2e-6 * (y.double().abs().T @ dqkv.double().abs()) + 1e-5
    assert ((dW.cpu().double() - want_dW).abs() <= tol_w).all()
    tol_b = 1e-4 * want_db.abs() + 2e-6 * dqkv.double().abs().sum(0) + 2e-4
    assert ((db.cpu().double() - want_db).abs() <= tol_b).all()
