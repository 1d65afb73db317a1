"""GPU parity: interval fusion kernels (LSTM, layer-norm over (T, d), MHSA + mean) against the
oracle. fp32, tolerance 1e-4 (north_star)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import selfgnn_oracle as O

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-4, 2e-5


def _params(d, rng, dev):
    p = O.init_fusion_params(d, rng)
    return p, {k: torch.from_numpy(v).to(dev) for k, v in p.items()}


@pytest.mark.parametrize("d,t,n", [(64, 3, 1000), (32, 1, 77), (128, 6, 301), (64, 16, 130), (48, 5, 64), (256, 2, 33)])
def test_lstm_vs_oracle(dev, d, t, n):
    from sa_gnn_amd import ops
    rng = np.random.default_rng(d * t)
    x = rng.standard_normal((n, t, d)).astype(np.float32)
    p, pd = _params(d, rng, dev)
    got = ops.lstm_fwd(torch.from_numpy(x).to(dev), pd["lstm_W"], pd["lstm_b"], 1.0)
    np.testing.assert_allclose(got.cpu().numpy(), O.basic_lstm(x, p["lstm_W"], p["lstm_b"], 1.0), rtol=RTOL, atol=ATOL)
    scale = ((rng.random((n, t, d)) < 0.5) * 2.0).astype(np.float32)
    got = ops.lstm_fwd(torch.from_numpy(x).to(dev), pd["lstm_W"], pd["lstm_b"], 1.0, drop_scale=torch.from_numpy(scale).to(dev))
    np.testing.assert_allclose(got.cpu().numpy(), O.basic_lstm(x, p["lstm_W"], p["lstm_b"], 1.0, scale), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("d,t,n", [(64, 3, 513), (32, 1, 5), (128, 12, 70), (48, 5, 9)])
def test_layernorm_vs_oracle(dev, d, t, n):
    from sa_gnn_amd import ops
    rng = np.random.default_rng(d + t)
    x = (rng.standard_normal((n, t, d)) * 2 + 0.5).astype(np.float32)
    p, pd = _params(d, rng, dev)
    xd = torch.from_numpy(x).to(dev)
    got = ops.layernorm_td(xd, pd["ln_gamma"], pd["ln_beta"])
    want = O.layer_norm_td(x, p["ln_gamma"], p["ln_beta"])
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    ops.layernorm_td(xd, pd["ln_gamma"], pd["ln_beta"], out=xd)          # in place
    np.testing.assert_allclose(xd.cpu().numpy(), want, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("d,heads,t,n", [(64, 16, 3, 700), (32, 16, 1, 50), (128, 16, 6, 90), (64, 4, 16, 40), (64, 16, 12, 257)])
def test_mhsa_mean_vs_oracle(dev, d, heads, t, n):
    from sa_gnn_amd import ops
    rng = np.random.default_rng(d + heads + t)
    x = rng.standard_normal((n, t, d)).astype(np.float32)
    p, pd = _params(d, rng, dev)
    got = ops.mhsa_mean(torch.from_numpy(x).to(dev), pd["Wq"], pd["bq"], pd["Wk"], pd["bk"], pd["Wv"], pd["bv"], heads)
    want = O.mhsa(x, p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], heads).mean(axis=1)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("d,t,n", [(64, 3, 900), (128, 6, 200), (32, 1, 100), (64, 16, 300)])
def test_interval_fusion_vs_oracle(dev, d, t, n):
    from sa_gnn_amd import ops
    rng = np.random.default_rng(d * 3 + t)
    x = rng.standard_normal((n, t, d)).astype(np.float32)
    p, pd = _params(d, rng, dev)
    got = ops.interval_fusion(torch.from_numpy(x).to(dev), pd, 16)
    np.testing.assert_allclose(got.cpu().numpy(), O.interval_fusion(x, p, 16), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("d", [32, 64])
def test_lstm_saturated_gates(dev, d):
    """Inputs large enough to saturate every gate (|pre-activation| up to a few hundred): the
    exp2-based sigmoid / tanh of the MFMA kernel must clamp to 0 / 1 / -1 like the oracle, no NaN."""
    from sa_gnn_amd import ops
    rng = np.random.default_rng(d)
    n, t = 257, 4
    x = (rng.standard_normal((n, t, d)) * 40.0).astype(np.float32)
    p, pd = _params(d, rng, dev)
    got = ops.lstm_fwd(torch.from_numpy(x).to(dev), pd["lstm_W"], pd["lstm_b"]).cpu().numpy()
    assert np.isfinite(got).all()
    with np.errstate(over="ignore"):
        want = O.basic_lstm(x, p["lstm_W"], p["lstm_b"], 1.0)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("d,n", [(64, 1000), (32, 333), (128, 77)])
def test_lstm_continuation_is_bit_identical(dev, d, n):
    """A sequence cut into consecutive calls (h0 / c0 in, cell state out) equals one call bit for
    bit — what the multi-GPU pipeline relies on when it runs the steps of the intervals that have
    arrived while the last exchange round is still in flight. x is a [t, n, d] slab viewed [n, t, d]."""
    from sa_gnn_amd import ops
    rng = np.random.default_rng(d + n)
    t, cut = 7, 3
    xs = torch.from_numpy(rng.standard_normal((t, n, d)).astype(np.float32)).to(dev)
    p, pd = _params(d, rng, dev)
    whole = ops.lstm_fwd(xs.permute(1, 0, 2), pd["lstm_W"], pd["lstm_b"])
    np.testing.assert_allclose(whole.cpu().numpy(), O.basic_lstm(xs.permute(1, 0, 2).cpu().numpy(), p["lstm_W"], p["lstm_b"], 1.0),
                               rtol=RTOL, atol=ATOL)
    h = torch.empty((n, t, d), device=dev)
    c = torch.empty((n, d), device=dev)
    ops.lstm_fwd(xs[:cut].permute(1, 0, 2), pd["lstm_W"], pd["lstm_b"], out=h[:, :cut, :], c_out=c)
    ops.lstm_fwd(xs[cut:].permute(1, 0, 2), pd["lstm_W"], pd["lstm_b"], out=h[:, cut:, :], h0=h[:, cut - 1, :], c0=c, c_out=c)
    assert torch.equal(h, whole)


@pytest.mark.parametrize("d", [32, 64])
@pytest.mark.parametrize("t", [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 32])
def test_attention_forms_every_interval_count(dev, d, t):
    """The attention kernel is specialised on the interval count (pair form for T <= 8, preloaded
    head-split form for 12 / 16, run-time form otherwise): each form, with and without the fused
    layer norm, against the oracle; node counts that leave ragged wave tiles."""
    from sa_gnn_amd import ops
    rng = np.random.default_rng(1000 * d + t)
    n = 4 * (32 // t) * 3 + 5
    x = rng.standard_normal((n, t, d)).astype(np.float32)
    p, pd = _params(d, rng, dev)
    xd = torch.from_numpy(x).to(dev)
    got = ops.mhsa_mean(xd, pd["Wq"], pd["bq"], pd["Wk"], pd["bk"], pd["Wv"], pd["bv"], 16)
    want = O.mhsa(x, p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], 16).mean(axis=1)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    got = ops.ln_mhsa_mean(xd, pd["ln_gamma"], pd["ln_beta"], pd["Wq"], pd["bq"], pd["Wk"], pd["bk"], pd["Wv"], pd["bv"], 16)
    y = O.layer_norm_td(x, p["ln_gamma"], p["ln_beta"])
    want = O.mhsa(y, p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], 16).mean(axis=1)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("d", [32, 64, 128])
def test_fusion_golden(dev, d):
    from sa_gnn_amd import ops
    g = np.load(os.path.join(GOLDEN, "oracle_tiny.npz"))
    pd = {k.split("/")[-1]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith(f"d{d}/p/")}
    uv = torch.from_numpy(g[f"d{d}/user_vector"]).to(dev)
    np.testing.assert_allclose(ops.lstm_fwd(uv, pd["lstm_W"], pd["lstm_b"]).cpu().numpy(), g[f"d{d}/lstm_user"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(ops.interval_fusion(uv, pd, 16).cpu().numpy(), g[f"d{d}/final_user"], rtol=RTOL, atol=ATOL)
    iv = torch.from_numpy(g[f"d{d}/item_vector"]).to(dev)
    np.testing.assert_allclose(ops.interval_fusion(iv, pd, 16).cpu().numpy(), g[f"d{d}/final_item"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("d,t", [(64, 4), (32, 16), (128, 3), (64, 16)])
def test_a_rows_result_does_not_depend_on_its_position_in_a_tile(dev, d, t):
    """The persistent kernels walk 96-row (LSTM) / 64-row (attention) tiles; the multi-GPU pipeline feeds them row
    shards and row chunks, so a node sits at a different tile position at every world size. Its h and its fused row
    must be BIT-identical wherever it sits (a hand-scheduled kernel whose per-tile instances were compiled to
    different instruction forms would break this: measured once with packed gate math)."""
    from sa_gnn_amd import ops
    from sa_gnn_amd.model import random_fusion_params
    n = 960
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.rand((n, t, d), generator=g, device=dev) * 0.06 - 0.03
    p = random_fusion_params(d, dev, 8)
    att = lambda h: ops.ln_mhsa_mean(h, p["ln_gamma"], p["ln_beta"], p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], 16)   # noqa: E731
    h = ops.lstm_fwd(x, p["lstm_W"], p["lstm_b"])
    f = att(h)
    for off in (1, 2, 15, 30, 96, 101):
        h2 = ops.lstm_fwd(x[off:].contiguous(), p["lstm_W"], p["lstm_b"])
        assert torch.equal(h2, h[off:]), f"LSTM rows move with the tile offset {off}"
        assert torch.equal(att(h2), f[off:]), f"attention rows move with the tile offset {off}"


@pytest.mark.parametrize("d,t,n", [(128, 6, 3_001), (128, 3, 70_001)])
def test_training_lstm_with_an_output_dropout_mask(dev, d, t, n):
    """sagnn_lstm_fwd_train_f32 with drop_scale (DropoutWrapper(output_keep_prob), model.py:137-140): the emitted h is
    h * mask, the recurrence runs on the un-dropped h. At d = 128 the recurrence travels through HBM (one launch per step),
    where h's slot now holds the dropped value: the next launch re-makes the un-dropped h_{t-1} from the saved cell state and
    o gate — the gates and cell states must be bit-identical to the call without a mask, on the matrix-core kernels."""
    from sa_gnn_amd import ops
    lib = ops._lib.load()
    rng = np.random.default_rng(d + t)
    x = torch.from_numpy(rng.standard_normal((n, t, d)).astype(np.float32)).to(dev)
    p = O.init_fusion_params(d, rng)
    W, b = torch.from_numpy(p["lstm_W"]).to(dev), torch.from_numpy(p["lstm_b"]).to(dev)
    drop = torch.from_numpy(((rng.random((n, t, d)) < 0.6) / 0.6).astype(np.float32)).to(dev)
    res = []
    for mask in (None, drop):
        h = torch.empty((n, t, d), device=dev)
        gates = torch.empty((n, t, 4 * d), device=dev)
        cell = torch.empty((n, t, d), device=dev)
        ops.range_redo_count(reset=True)
        ops.check(lib.sagnn_lstm_fwd_train_f32(x.data_ptr(), t * d, d, n, t, d, W.data_ptr(), b.data_ptr(), 1.0,
                                               None if mask is None else mask.data_ptr(), h.data_ptr(), t * d, gates.data_ptr(),
                                               cell.data_ptr(), ops._stream()))
        assert ops.range_redo_count() == 0
        res.append((h, gates, cell))
    (h0, g0, c0), (h1, g1, c1) = res
    assert torch.equal(g0, g1) and torch.equal(c0, c1)
    assert torch.equal(h1, h0 * drop)
    want = O.basic_lstm(x.cpu().numpy(), p["lstm_W"], p["lstm_b"], 1.0)
    np.testing.assert_allclose(h0.cpu().numpy(), want, rtol=1e-4, atol=2e-5)
