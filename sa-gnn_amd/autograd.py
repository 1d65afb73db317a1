"""torch.autograd bindings for the hot path (reference: the gradients tf.train.AdamOptimizer
.minimize builds for model.py:118-129, model.py:250). Forward and backward both run in
libsagnn.so; torch only carries the graph."""
from __future__ import annotations

import torch

from . import ops


class GnnIntervalFn(torch.autograd.Function):
    """(uEmbed[k], iEmbed[k]) -> (user_k, item_k): L propagation layers with residuals and add_n.
    Saves one activation mask per layer and direction (d/4 bytes per row) instead of the layer
    outputs: the backward pass needs only the slopes."""

    @staticmethod
    def forward(ctx, u0, i0, plan_user, plan_item, n_layers, leaky):
        U, I, d = plan_user.n_rows, plan_item.n_rows, int(u0.shape[1])
        dev = u0.device
        user_out = torch.empty((U, d), dtype=torch.float32, device=dev)
        item_out = torch.empty((I, d), dtype=torch.float32, device=dev)
        mask_u = torch.empty((n_layers, U, d // 4), dtype=torch.uint8, device=dev)
        mask_i = torch.empty((n_layers, I, d // 4), dtype=torch.uint8, device=dev)
        ops.gnn_interval(plan_user, plan_item, u0.detach().contiguous(), i0.detach().contiguous(), n_layers,
                         leaky, user_out, item_out, mask_u=mask_u, mask_i=mask_i)
        ctx.save_for_backward(mask_u, mask_i)
        ctx.plans = (plan_user, plan_item)
        ctx.cfg = (n_layers, leaky)
        return user_out, item_out

    @staticmethod
    def backward(ctx, g_user, g_item):
        mask_u, mask_i = ctx.saved_tensors
        plan_user, plan_item = ctx.plans
        n_layers, leaky = ctx.cfg
        U, I, d = plan_user.n_rows, plan_item.n_rows, mask_u.shape[2] * 4
        if g_user is None:
            g_user = torch.zeros((U, d), dtype=torch.float32, device=mask_u.device)
        if g_item is None:
            g_item = torch.zeros((I, d), dtype=torch.float32, device=mask_u.device)
        du, di = ops.gnn_interval_bwd(plan_user, plan_item, g_user.contiguous(), g_item.contiguous(), n_layers,
                                      leaky, mask_u, mask_i)
        return du, di, None, None, None, None


def gnn_interval(u0, i0, plan_user, plan_item, n_layers: int, leaky: float):
    return GnnIntervalFn.apply(u0, i0, plan_user, plan_item, n_layers, leaky)


class IntervalFusionFn(torch.autograd.Function):
    """x [n, t, d] (any node/interval strides) + fusion parameters -> out [n, d]
    (reference model.py:135-155), differentiable in x and every parameter.

    Forward = LSTM (saving gate activations and cell states) + fused layer-norm/attention kernel.
    Backward recomputes y = LN(h) and Q|K|V with the forward kernels, then:
      attention backward (per node) -> dQ|dK|dV -> dW_qkv / db_qkv (dense tn) and dy (dense nn)
      -> layer-norm backward -> BPTT: per step an element-wise gate backward, dW_lstm += [x_t|h_{t-1}]^T
      dgates (dense tn) and d[x_t | h_{t-1}] = dgates @ W^T (dense nn).
    Parameter order: lstm_W, lstm_b, ln_gamma, ln_beta, Wq, bq, Wk, bk, Wv, bv."""

    @staticmethod
    def forward(ctx, x, lstm_W, lstm_b, ln_gamma, ln_beta, Wq, bq, Wk, bk, Wv, bv, heads, drop_scale):
        lib = ops._lib.load()
        n, t, d, ld_n, ld_t = ops._ntd("x", x)
        dev = x.device
        h = torch.empty((n, t, d), dtype=torch.float32, device=dev)
        gates = torch.empty((n, t, 4 * d), dtype=torch.float32, device=dev)
        cell = torch.empty((n, t, d), dtype=torch.float32, device=dev)
        # h is stored un-dropped (it is also the recurrent operand of the backward pass); the
        # DropoutWrapper scaling of the emitted output is a separate element-wise pass
        ops.check(lib.sagnn_lstm_fwd_train_f32(
            x.data_ptr(), ld_n, ld_t, n, t, d, ops._vec("lstm_W", lstm_W.detach(), 8 * d * d),
            ops._vec("lstm_b", lstm_b.detach(), 4 * d), 1.0, None, h.data_ptr(), t * d,
            gates.data_ptr(), cell.data_ptr(), ops._stream()))
        h_emit = h if drop_scale is None else ops.mul(h, drop_scale.contiguous())
        y = ops.layernorm_td(h_emit, ln_gamma.detach(), ln_beta.detach())
        out = ops.mhsa_mean(y, Wq.detach(), bq.detach(), Wk.detach(), bk.detach(), Wv.detach(), bv.detach(), heads)
        ctx.save_for_backward(x, lstm_W, ln_gamma, ln_beta, Wq, bq, Wk, bk, Wv, bv, h, gates, cell,
                              drop_scale if drop_scale is not None else torch.empty(0, device=dev))
        ctx.heads = heads
        ctx.has_drop = drop_scale is not None
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = ops._lib.load()
        (x, lstm_W, ln_gamma, ln_beta, Wq, bq, Wk, bk, Wv, bv, h, gates, cell, drop) = ctx.saved_tensors
        drop = drop if ctx.has_drop else None
        heads = ctx.heads
        n, t, d, ld_n, ld_t = ops._ntd("x", x)
        dev = x.device
        st = ops._stream()
        g_out = g_out.contiguous()
        # ---- recompute y and Q|K|V ---------------------------------------------------------
        h_emit = h if drop is None else ops.mul(h, drop.contiguous())
        y = ops.layernorm_td(h_emit, ln_gamma.detach(), ln_beta.detach())                # [n, t, d]
        Wqkv = torch.cat([Wq, Wk, Wv], dim=1).detach().contiguous()                      # [d, 3d]
        bqkv = torch.cat([bq, bk, bv]).detach().contiguous()
        y2 = y.view(n * t, d)
        qkv = ops.dense_nn(y2, Wqkv, bqkv)                                               # [n*t, 3d]
        # ---- attention backward, in place: qkv -> dQ|dK|dV -----------------------------------
        ops.check(lib.sagnn_attn_bwd_f32(qkv.data_ptr(), g_out.data_ptr(), d, n, t, d, heads, st))
        dWqkv = torch.zeros((d, 3 * d), dtype=torch.float32, device=dev)
        dbqkv = torch.zeros(3 * d, dtype=torch.float32, device=dev)
        ops.dense_tn(y2, qkv, dWqkv, dbqkv)
        dy = ops.dense_nn(qkv, Wqkv.t().contiguous(), None, out=y2)                      # reuses y's storage
        # ---- layer norm backward (in place on dy) ----------------------------------------------
        dgamma = torch.zeros(d, dtype=torch.float32, device=dev)
        dbeta = torch.zeros(d, dtype=torch.float32, device=dev)
        dh = dy.view(n, t, d)
        ops.check(lib.sagnn_layernorm_td_bwd_f32(h_emit.data_ptr(), t * d, dh.data_ptr(), t * d, n, t, d,
                                                 ops._vec("gamma", ln_gamma.detach(), d), 1e-12, dh.data_ptr(),
                                                 t * d, dgamma.data_ptr(), dbeta.data_ptr(), st))
        # ---- BPTT ----------------------------------------------------------------------------------
        Wd = lstm_W.detach()
        WxT = Wd[:d].t().contiguous()                                                    # [4d, d]
        WhT = Wd[d:].t().contiguous()
        dW = torch.zeros((2 * d, 4 * d), dtype=torch.float32, device=dev)
        db = torch.zeros(4 * d, dtype=torch.float32, device=dev)
        dx = torch.empty((n, t, d), dtype=torch.float32, device=dev)
        dgates = torch.empty((n, 4 * d), dtype=torch.float32, device=dev)
        dc = [torch.empty((n, d), dtype=torch.float32, device=dev) for _ in range(2)]
        dh_rec = torch.empty((n, d), dtype=torch.float32, device=dev)
        h_state = h                                   # un-dropped: the recurrent operand
        for ts in range(t - 1, -1, -1):
            last = ts == t - 1
            ops.check(lib.sagnn_lstm_bwd_step_f32(
                gates.data_ptr(), cell.data_ptr(), dh.data_ptr(), t * d, ops._ptr(drop),
                None if last else dh_rec.data_ptr(), d, None if last else dc[(ts + 1) & 1].data_ptr(),
                dgates.data_ptr(), dc[ts & 1].data_ptr(), n, t, d, ts, st))
            x_t = x[:, ts, :]
            ops.dense_tn(x_t, dgates, dW[:d], db)
            if ts > 0:
                ops.dense_tn(h_state[:, ts - 1, :], dgates, dW[d:], None)
                ops.dense_nn(dgates, WhT, None, out=dh_rec)
            ops.dense_nn(dgates, WxT, None, out=dx[:, ts, :])
        dWq, dWk, dWv = (dWqkv[:, i * d:(i + 1) * d].contiguous() for i in range(3))
        dbq, dbk, dbv = (dbqkv[i * d:(i + 1) * d].contiguous() for i in range(3))
        return dx, dW, db, dgamma, dbeta, dWq, dbq, dWk, dbk, dWv, dbv, None, None


def interval_fusion(x, p: dict, heads: int, drop_scale=None):
    """Differentiable interval fusion; p as in ops.interval_fusion."""
    return IntervalFusionFn.apply(x, p["lstm_W"], p["lstm_b"], p["ln_gamma"], p["ln_beta"], p["Wq"], p["bq"],
                                  p["Wk"], p["bk"], p["Wv"], p["bv"], heads, drop_scale)
