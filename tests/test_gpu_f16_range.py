"""The exponent window of the f16 x 2 engine (sa-gnn_amd/csrc/f16_split.h, RANGE), tested where it bites.

The default engine evaluates the GEMM-shaped fusion stages on two-piece f16 operands: exact to 2^-23 |v| between an
absolute floor of 2^-37 and |v| < 32768. Forward operands have a natural scale (layer-normed rows, embedding sums),
so the kernels only CHECK the window and redo a tile in fp32 when an operand leaves it; gradients have none, so the
attention-backward tail scales every gradient row by an exact power of two. These tests put rows of very different
scales into one call and judge every row at a tolerance proportional to ITS OWN scale — one that a kernel which
flushed the small rows to zero would fail (the oracle's output depends on them linearly: biases are TF's initial
zeros here). Reference: tf.gradients through Utils/attention.py:66-72 and model.py:135-155, 244-250 (fp32 MatMul).
Both engines run every case; the float64 oracle is the judge."""
import numpy as np
import pytest
import torch

from oracle import selfgnn_oracle as O

pytestmark = pytest.mark.gpu

ENGINES = ["f16x2", "f32"]
SCALES = [1.0, 1e-6, 1e-9, 1e-12]


def _row_scales(rows, rng, pattern):
    """One scale per row. blocks: contiguous quarters; mixed: every row its own draw (neighbours in a 32-row chunk
    differ by up to 1e12); rising / falling: the kernel's running scale has to follow / must not forget;
    tiny: nothing of ordinary size anywhere in the call; needle: one row of ordinary size in a field of 1e-9."""
    s = np.empty(rows)
    if pattern == "blocks":
        q = (np.arange(rows) * len(SCALES)) // rows
        s[:] = np.asarray(SCALES)[q]
    elif pattern == "mixed":
        s[:] = rng.choice(SCALES, size=rows)
    elif pattern == "rising":
        s[:] = 10.0 ** np.linspace(-30, 20, rows)
    elif pattern == "falling":
        s[:] = 10.0 ** np.linspace(20, -30, rows)
    elif pattern == "tiny":
        s[:] = 1e-12 * 10.0 ** rng.uniform(-3, 0, size=rows)
    elif pattern == "needle":
        s[:] = 1e-9
        s[rows // 3] = 1.0
    else:
        raise ValueError(pattern)
    return s


def _per_row_ok(got, want, what, rtol=1e-4, atol=0.0):
    """|err| <= rtol * max|row| (+ atol[row]) for every row: a row's own scale, not the tensor's. With atol = 0 a
    kernel that returned zeros for a small row fails."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    g2, w2 = got.reshape(got.shape[0], -1), want.reshape(want.shape[0], -1)
    assert np.isfinite(g2).all(), f"{what}: non-finite values"
    scale = np.abs(w2).max(axis=1)
    err = np.abs(g2 - w2).max(axis=1)
    over = err / (rtol * scale + atol + 1e-300)
    worst = int(np.argmax(over))
    assert over[worst] <= 1.0, (f"{what}: {int((over > 1).sum())} of {len(over)} rows off; worst row {worst}: err {err[worst]:.3e}, "
                                f"row max {scale[worst]:.3e}")


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("pattern", ["blocks", "mixed", "rising", "falling", "tiny", "needle"])
@pytest.mark.parametrize("rows,d", [(40_007, 64), (1_000, 64), (70_001, 32)])
def test_attn_bwd_tail_rows_of_any_scale(dev, rows, d, pattern, engine):
    """sagnn_attn_bwd_tail_f32 with gradient rows scaled by 1, 1e-6, 1e-9, 1e-12 (and 1e-30 .. 1e20) in ONE call.
    dy is judged per row at 1e-4 of the row's own largest entry; dW and db against the sum of the magnitudes of
    their terms (the measure of an fp32 sum) plus 1e-4 of their own value."""
    from sa_gnn_amd import _lib, ops
    lib = _lib.load()
    rng = np.random.default_rng(rows + d + len(pattern))
    y = rng.standard_normal((rows, d)).astype(np.float32)
    sc = _row_scales(rows, rng, pattern)
    dqkv = (rng.standard_normal((rows, 3 * d)) * sc[:, None]).astype(np.float32)
    W = (rng.standard_normal((d, 3 * d)) / d ** 0.5).astype(np.float32)
    yd, gd, Wd = (torch.from_numpy(a).to(dev) for a in (y, dqkv, W))
    dW = torch.zeros((d, 3 * d), device=dev)
    db = torch.zeros(3 * d, device=dev)
    ops.range_redo_count(reset=True)
    with ops.engine(engine):
        ops.check(lib.sagnn_attn_bwd_tail_f32(yd.data_ptr(), gd.data_ptr(), rows, d, Wd.data_ptr(), dW.data_ptr(),
                                              db.data_ptr(), None))
    redo = ops.range_redo_count()
    n_chunks = (rows + 31) // 32
    if engine == "f16x2" and pattern in ("blocks", "tiny", "falling"):
        # scales that the running exponent follows without help: the matrix cores did (nearly) all of it. "mixed" and
        # "needle" send the first chunk that meets a row 1e6 x larger than anything before it through fp32, "rising"
        # (every chunk 1e10 x the previous one of its block) all but each block's first chunk.
        assert redo <= max(2, n_chunks // 50), (redo, n_chunks)
    y64, g64, W64 = y.astype(np.float64), dqkv.astype(np.float64), W.astype(np.float64)
    _per_row_ok(yd.cpu().numpy(), g64 @ W64.T, f"dy[{pattern}]")
    want_dW, mag_dW = y64.T @ g64, np.abs(y64).T @ np.abs(g64)
    err = np.abs(dW.cpu().numpy().astype(np.float64) - want_dW)
    tol = 1e-4 * np.abs(want_dW) + 2e-6 * mag_dW
    assert (err <= tol).all(), f"dW[{pattern}]: worst err / sum|terms| = {(err / mag_dW).max():.3e}"
    want_db, mag_db = g64.sum(0), np.abs(g64).sum(0)
    err = np.abs(db.cpu().numpy().astype(np.float64) - want_db)
    assert (err <= 1e-4 * np.abs(want_db) + 2e-6 * mag_db).all(), f"db[{pattern}]: worst err / sum|terms| = {(err / mag_db).max():.3e}"
    assert mag_dW.min() > 0 and (2e-6 * mag_dW < 0.01 * mag_dW).all()


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("d,t,n", [(64, 2, 20_011), (64, 3, 9_001), (32, 4, 12_345)])
def test_fusion_backward_dx_per_node_at_any_gradient_scale(dev, d, t, n, engine):
    """The whole fusion backward (attention-backward front and tail, layer-norm backward, BPTT) with the upstream
    gradient of node i scaled by 1, 1e-6, 1e-9 or 1e-12 (reference: --ssl_reg 1e-6 next to hinge rows, gowalla.sh:1).
    The fusion is independent per node, so dx[i] scales with its node's factor: every node's [t, d] block is judged at
    1e-4 of its own largest entry against float64 autograd."""
    from sa_gnn_amd import autograd as ag
    from sa_gnn_amd import ops
    rng = np.random.default_rng(d * 10 + t)
    x = rng.standard_normal((n, t, d)).astype(np.float32)
    p = O.init_fusion_params(d, rng)
    sc = rng.choice(SCALES, size=n)
    gout = (rng.standard_normal((n, d)) * sc[:, None]).astype(np.float32)
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    (O.torch_interval_fusion(tx, tp, 16) * torch.tensor(gout, dtype=torch.float64)).sum().backward()
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    pd = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in p.items()}
    with ops.engine(engine):
        ag.interval_fusion(xd, pd, 16).backward(torch.from_numpy(gout).to(dev))
    _per_row_ok(xd.grad.cpu().numpy(), tx.grad.numpy(), "dx")
    # parameter gradients: sums over all nodes, dominated by the ordinary-sized ones
    for k in p:
        a, b = pd[k].grad.cpu().numpy().astype(np.float64), tp[k].grad.numpy()
        floor = 5e-6 * max(1.0, np.sqrt(n * t / 1000.0))      # accumulation noise of sums that cancel to ~0 (test_gpu_backward)
        assert (np.abs(a - b) <= 1e-4 * np.abs(b) + max(2e-5 * np.abs(b).max(), floor)).all(), k


def _zero_bias_params(d, rng, dev):
    """TF's own initial values for the additive terms (BasicLSTMCell bias zeros, layer_norm beta zeros, dense biases
    zeros): with them the fused output of a node is LINEAR in a small input, so small inputs stay visible."""
    p = O.init_fusion_params(d, rng)
    for k in ("lstm_b", "ln_beta", "bq", "bk", "bv"):
        p[k] = np.zeros_like(p[k])
    return p, {k: torch.from_numpy(v).to(dev) for k, v in p.items()}


# The gate non-linearities are exp2 / rcp forms (sigmoid = 1 / (1 + 2^t), tanh = 1 - 2 / (1 + 2^t)): accurate to a few
# 1e-7 ABSOLUTE on values bounded by one, under every engine. TF's tanh is relatively accurate near zero; here a row
# whose pre-activations are all below 1e-3 keeps that absolute accuracy on h (h itself is then below 1e-3), unless it is
# so small that the f16 window check sends its tile to the kernel's fp32 pass, which evaluates tanhf.
GATE_ABS = 5e-7


@pytest.mark.parametrize("pattern", ["blocks", "mixed", "rising", "tiny"])
@pytest.mark.parametrize("d,t,n", [(64, 5, 20_011), (32, 6, 12_345), (64, 1, 5_003)])
def test_lstm_weight_gradient_pass_at_any_gradient_scale(dev, d, t, n, pattern):
    """sagnn_lstm_bwd_ws_f32: the BPTT launch without its dW product + the weight gradient as a second pass over the
    stored gate gradients on the f16 x 2 engine (the attention tail's kernel in its LSTM form: [x_s | h_{s-1}] against
    dG_s, rows of dG scaled by exact powers of two). Nodes whose upstream gradient is 1, 1e-6, 1e-9, 1e-12 in one call:
    dx and db are the one-launch kernel's bit for bit (same code), the gate gradients in the scratch are what the float64
    product is taken from, and dW must match it to 1e-6 of the sum of its terms' magnitudes. Every column of dW sums all
    rows, so the pattern that a kernel flushing small rows cannot pass is 'tiny': ALL rows are 1e-12 .. 1e-15 there."""
    from sa_gnn_amd import _lib, ops
    from sa_gnn_amd.model import random_fusion_params
    lib = _lib.load()
    rng = np.random.default_rng(d + t + len(pattern))
    g = torch.Generator(device="cpu").manual_seed(d * 3 + t)
    x = (torch.rand((n, t, d), generator=g) * 2 - 1).to(dev)
    p = random_fusion_params(d, dev, 11)
    h = torch.empty((n, t, d), device=dev)
    gates = torch.empty((n, t, 4 * d), device=dev)
    cell = torch.empty((n, t, d), device=dev)
    ops.check(lib.sagnn_lstm_fwd_train_f32(x.data_ptr(), t * d, d, n, t, d, p["lstm_W"].data_ptr(), p["lstm_b"].data_ptr(), 1.0, None,
                                           h.data_ptr(), t * d, gates.data_ptr(), cell.data_ptr(), None))
    scale = torch.from_numpy(_row_scales(n, rng, pattern).astype(np.float32)).to(dev)
    dh = torch.randn((n, t, d), generator=g).to(dev) * scale[:, None, None]
    outs = []
    for use_ws in (False, True):
        dx = torch.empty((n, t, d), device=dev)
        dW = torch.zeros((2 * d, 4 * d), device=dev)
        db = torch.zeros(4 * d, device=dev)
        nbytes = int(lib.sagnn_lstm_bwd_workspace_bytes(n, t, d))
        assert nbytes == n * t * 4 * d * 4
        ws = torch.zeros(nbytes // 4, device=dev) if use_ws else None
        ops.range_redo_count(reset=True)
        ops.check(lib.sagnn_lstm_bwd_ws_f32(x.data_ptr(), t * d, d, h.data_ptr(), gates.data_ptr(), cell.data_ptr(), dh.data_ptr(), t * d,
                                            None, p["lstm_W"].data_ptr(), dx.data_ptr(), dW.data_ptr(), db.data_ptr(), n, t, d,
                                            ops._ptr(ws), nbytes if use_ws else 0, None))
        outs.append((dx, dW, db, ws))
    (dx0, dW0, db0, _), (dx1, dW1, db1, ws) = outs
    assert torch.equal(dx0, dx1)
    torch.testing.assert_close(db1, db0, rtol=1e-5, atol=1e-6 * float(db0.abs().max()))      # float atomics: order differs
    dG = ws.view(t, n, 4 * d).double()                                                   # time-major
    xh = torch.cat([x.permute(1, 0, 2), torch.cat([torch.zeros((1, n, d), device=dev), h.permute(1, 0, 2)[:-1]])], dim=2).double()
    want = torch.einsum("tnk,tng->kg", xh, dG)
    mag = torch.einsum("tnk,tng->kg", xh.abs(), dG.abs())
    assert float(mag.max()) > 0
    for name, got in (("second pass", dW1), ("one launch", dW0)):
        err = (got.double() - want).abs()
        assert bool((err <= 1e-6 * mag + 1e-30).all()), f"{name} [{pattern}]: worst {float((err / (mag + 1e-300)).max()):.3e} of sum |terms|"
    # rows are met step by step (time-major): 'blocks' and 'rising' start every step 1e12 .. 1e50 above where the previous
    # one ended, and the chunks at such a jump take the fp32 path by design (the scale follows by 2^16 per chunk)
    if pattern in ("mixed", "tiny"):
        assert ops.range_redo_count() <= max(2, n * t // 32 // 50), "ordinary gradient rows must stay on the matrix cores"


@pytest.mark.parametrize("d,t,n", [(64, 6, 9_001), (32, 4, 5_003), (64, 4, 19)])
def test_lstm_weight_gradient_pass_with_an_output_dropout_mask(dev, d, t, n):
    """The second-pass form under DropoutWrapper(output_keep_prob) (reference model.py:141-144): the mask scales the gradient
    arriving at the emitted h inside the BPTT launch, the weight-gradient pass reads the UN-dropped h as the one-launch kernel
    does. dx bit for bit, dW / db against the one-launch kernel."""
    from sa_gnn_amd import _lib, ops
    from sa_gnn_amd.model import random_fusion_params
    lib = _lib.load()
    g = torch.Generator(device="cpu").manual_seed(d + t + n)
    x = (torch.rand((n, t, d), generator=g) * 2 - 1).to(dev)
    p = random_fusion_params(d, dev, 5)
    drop = ((torch.rand((n, t, d), generator=g) < 0.5).float() * 2.0).to(dev)
    h = torch.empty((n, t, d), device=dev)
    gates = torch.empty((n, t, 4 * d), device=dev)
    cell = torch.empty((n, t, d), device=dev)
    ops.check(lib.sagnn_lstm_fwd_train_f32(x.data_ptr(), t * d, d, n, t, d, p["lstm_W"].data_ptr(), p["lstm_b"].data_ptr(), 1.0, None,
                                           h.data_ptr(), t * d, gates.data_ptr(), cell.data_ptr(), None))
    dh = torch.randn((n, t, d), generator=g).to(dev)
    outs = []
    for use_ws in (False, True):
        dx, dW, db = torch.empty((n, t, d), device=dev), torch.zeros((2 * d, 4 * d), device=dev), torch.zeros(4 * d, device=dev)
        nbytes = int(lib.sagnn_lstm_bwd_workspace_bytes(n, t, d))
        ws = torch.empty(nbytes // 4, device=dev) if use_ws else None
        ops.check(lib.sagnn_lstm_bwd_ws_f32(x.data_ptr(), t * d, d, h.data_ptr(), gates.data_ptr(), cell.data_ptr(), dh.data_ptr(), t * d,
                                            drop.data_ptr(), p["lstm_W"].data_ptr(), dx.data_ptr(), dW.data_ptr(), db.data_ptr(), n, t, d,
                                            ops._ptr(ws), nbytes if use_ws else 0, None))
        outs.append((dx, dW, db))
    assert torch.equal(outs[0][0], outs[1][0])
    dG = ws.view(t, n, 4 * d).double().abs()                  # what the second pass read: the size of the sums' terms
    xh = torch.cat([x.permute(1, 0, 2), torch.cat([torch.zeros((1, n, d), device=dev), h.permute(1, 0, 2)[:-1]])], dim=2).double().abs()
    mag = torch.einsum("tnk,tng->kg", xh, dG)
    assert bool(((outs[1][1] - outs[0][1]).abs().double() <= 2e-6 * mag + 1e-30).all())
    assert bool(((outs[1][2] - outs[0][2]).abs().double() <= 2e-6 * dG.sum((0, 1)) + 1e-30).all())
    with pytest.raises(_lib.SagnnError):                      # a scratch that is too small is an error, not a silent fallback
        ops.check(lib.sagnn_lstm_bwd_ws_f32(x.data_ptr(), t * d, d, h.data_ptr(), gates.data_ptr(), cell.data_ptr(), dh.data_ptr(), t * d,
                                            drop.data_ptr(), p["lstm_W"].data_ptr(), dx.data_ptr(), dW.data_ptr(), db.data_ptr(), n, t, d,
                                            ws.data_ptr(), 16, None))


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("d,t,n", [(64, 3, 2_000), (64, 16, 700), (32, 4, 1_500), (128, 3, 900), (64, 2, 40_003)])
def test_lstm_small_inputs_keep_their_own_accuracy(dev, d, t, n, engine):
    """x rows of 1e-4 ... 1e-30 next to ordinary rows. With zero biases h ~ 0.25 W_j x: a node's output is as small as
    its input. Nodes of 1e-7 and below leave the window of the f16 split: the default engine re-evaluates their tiles
    in fp32 and they are judged at 1e-4 of their own largest entry, NO absolute floor (zeros fail). Nodes of 1e-4 / 1e-5
    stay on the fast path and are judged at 1e-4 of their own scale + the absolute floor of the gate formulas (5e-7,
    still 4 to 40 times below their outputs). The exact-fp32 engine shares that floor and has no fp32-redo for tiny rows
    (its products are fp32 already), so for it every node carries the floor."""
    from sa_gnn_amd import ops
    if engine == "f32" and d == 128:
        pytest.skip("the f32-MFMA engine covers d = 32 / 64")
    rng = np.random.default_rng(7 * d + t)
    x = rng.standard_normal((n, t, d)).astype(np.float32)
    sc = rng.choice([1.0, 1.0, 1e-4, 1e-5, 1e-7, 1e-9, 1e-12, 1e-30], size=n)
    x = (x * sc[:, None, None]).astype(np.float32)
    p, pd = _zero_bias_params(d, rng, dev)
    with ops.engine(engine):
        got = ops.lstm_fwd(torch.from_numpy(x).to(dev), pd["lstm_W"], pd["lstm_b"], 1.0)
    want = O.basic_lstm(x.astype(np.float64), p["lstm_W"].astype(np.float64), p["lstm_b"].astype(np.float64), 1.0)
    assert 0 < np.abs(want[sc == 1e-30]).max() < 1e-28
    assert np.abs(want[sc == 1e-5]).reshape(-1, t * d).max(axis=1).min() > 4 * GATE_ABS     # the floor leaves these rows teeth
    tiny = sc <= 1e-7
    floor = np.where(tiny & (engine == "f16x2"), 0.0, GATE_ABS)
    _per_row_ok(got.cpu().numpy(), want, "lstm_fwd", atol=floor)


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("d,t,n", [(64, 3, 2_000), (64, 16, 700), (32, 4, 1_500), (128, 6, 900), (64, 2, 40_003)])
def test_attention_small_inputs_keep_their_own_accuracy(dev, d, t, n, engine):
    """Nodes whose [t, d] block is 1e-9 / 1e-12 / 1e-20 of the others: layer_norm's eps = 1e-12 stops normalising
    them (y ~ 1e6 x), so their Q|K|V and their fused output are tiny; judged per node at 1e-4 of the node's own scale."""
    from sa_gnn_amd import ops
    if engine == "f32" and d == 128:
        pytest.skip("the f32-MFMA engine covers d = 32 / 64")
    rng = np.random.default_rng(9 * d + t)
    sc = rng.choice([1.0, 1.0, 1e-9, 1e-12, 1e-20], size=n)
    x = (rng.standard_normal((n, t, d)) * sc[:, None, None]).astype(np.float32)
    p, pd = _zero_bias_params(d, rng, dev)
    with ops.engine(engine):
        got = ops.ln_mhsa_mean(torch.from_numpy(x).to(dev), pd["ln_gamma"], pd["ln_beta"], pd["Wq"], pd["bq"], pd["Wk"],
                               pd["bk"], pd["Wv"], pd["bv"], 16)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    y = O.layer_norm_td(x.astype(np.float64), p64["ln_gamma"], p64["ln_beta"])
    want = O.mhsa(y, p64["Wq"], p64["bq"], p64["Wk"], p64["bk"], p64["Wv"], p64["bv"], 16).mean(axis=1)
    assert 0 < np.abs(want[sc == 1e-20]).max() < 1e-12
    _per_row_ok(got.cpu().numpy(), want, "ln_mhsa_mean")


@pytest.mark.parametrize("d", [64, 32])
def test_lstm_redo_starts_from_the_callers_state_when_updated_in_place(dev, d):
    """A sequence cut into two chained calls with the cell state updated IN PLACE (c0 is c_out, as
    parallel.RoundFusion.lstm_round passes it), the second call holding an x beyond the split's window: the kernel's
    fp32 redo of that tile must start from the state the caller passed, not from what the fast pass left there."""
    from sa_gnn_amd import ops
    rng = np.random.default_rng(d)
    n, t1, t2 = 700, 2, 3
    x = rng.standard_normal((n, t1 + t2, d)).astype(np.float32)
    x[100, t1 + 1, 5] = 7e4            # second call, tile 1
    x[300, t1, :] *= 1e-25             # second call, another tile: below the window
    p = O.init_fusion_params(d, rng)
    pd = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}
    xd = torch.from_numpy(x).to(dev)
    h = torch.empty((n, t1 + t2, d), device=dev)
    c = torch.zeros((n, d), device=dev)
    ops.lstm_fwd(xd[:, :t1], pd["lstm_W"], pd["lstm_b"], 1.0, out=h[:, :t1], c_out=c)
    ops.lstm_fwd(xd[:, t1:], pd["lstm_W"], pd["lstm_b"], 1.0, out=h[:, t1:], h0=h[:, t1 - 1], c0=c, c_out=c)
    want = O.basic_lstm(x.astype(np.float64), p["lstm_W"].astype(np.float64), p["lstm_b"].astype(np.float64), 1.0)
    got = h.cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - want)
    assert (err <= 1e-4 * np.abs(want) + 2e-5).all(), f"worst {err.max():.3e} at node {int(np.argwhere(err == err.max())[0][0])}"
    assert torch.isfinite(c).all()


@pytest.mark.parametrize("v", [32784.0, 49168.0, 32768.0, 65504.0, -40000.0])
def test_top_binade_ties_take_the_fp32_path(dev, v):
    """|v| in [32768, 65504] fits an f16 head but its residual can reach 16, and 16 * 4096 overflows the scaled
    piece (32784 -> head 32768, tail Inf): the kernels redo such tiles in fp32 from 32768 up."""
    from sa_gnn_amd import ops
    d, t, n = 64, 2, 300
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, t, d)).astype(np.float32)
    x[10, 0, 3] = v
    x[200, 1, 60] = v
    p = O.init_fusion_params(d, rng)
    p["lstm_W"] = (p["lstm_W"] * 1e-4).astype(np.float32)       # gates stay off saturation: the big input shows in h
    pd = {k: torch.from_numpy(a).to(dev) for k, a in p.items()}
    got = ops.lstm_fwd(torch.from_numpy(x).to(dev), pd["lstm_W"], pd["lstm_b"], 1.0).cpu().numpy()
    want = O.basic_lstm(x.astype(np.float64), p["lstm_W"].astype(np.float64), p["lstm_b"].astype(np.float64), 1.0)
    assert np.isfinite(got).all()
    assert (np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-6).all()
