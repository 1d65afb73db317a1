"""torch.autograd bindings for the hot path (reference: the gradients tf.train.AdamOptimizer
.minimize builds for model.py:118-129, model.py:250). Forward and backward both run in
libsagnn.so; torch only carries the graph."""
from __future__ import annotations

import torch

from . import ops

# Host-side A/B switches (tests flip them): False routes the backward through the per-step / per-product entries
# that the fused entries (attention-backward front + tail, one-launch BPTT) replaced.
FUSED_ATTN_BWD = True
FUSED_BPTT = True
DEFERRED_DW_LIMIT = 32 << 30   # bytes of gate gradients ([t, n, 4d]) a BPTT may keep for the separate weight-gradient pass
SPLIT_DW = True                # one-launch BPTT (d = 32 / 64): weight gradient as a second pass on the f16 x 2 engine ...
SPLIT_DW_MIN_T = 4              # ... from this many steps on (measured: T = 16 -13 %, T = 6 -8 %, T = 2 / 3 nothing: the gate gradients' HBM round trip)


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


class GnnStackFn(torch.autograd.Function):
    """(uEmbed [T, U, d], iEmbed [T, I, d]) -> (user slab [T, U, d], item slab [T, I, d]): the whole loop of
    model.py:118-129 as ONE autograd node. Interval outputs are written straight into the slabs the fusion reads
    as [N, T, d] views (no torch.stack copy). With an ops.SpmmBatch the loop over k happens INSIDE the launches
    (one per layer, sagnn_gnn_stack_f32 / _bwd_f32); with plan lists every interval takes its own 2 L launches
    (sagnn_gnn_interval_ex_f32: graphs whose T-fold scratch would not fit)."""

    @staticmethod
    def forward(ctx, u_embed, i_embed, plans_user, plans_item, n_layers, leaky):
        T, U, d = u_embed.shape
        I = i_embed.shape[1]
        dev = u_embed.device
        ue, ie = u_embed.detach(), i_embed.detach()
        out_u = torch.empty((T, U, d), dtype=torch.float32, device=dev)
        out_i = torch.empty((T, I, d), dtype=torch.float32, device=dev)
        mask_u = torch.empty((T, n_layers, U, d // 4), dtype=torch.uint8, device=dev)
        mask_i = torch.empty((T, n_layers, I, d // 4), dtype=torch.uint8, device=dev)
        if isinstance(plans_user, ops.SpmmBatch):
            ops.gnn_stack(plans_user, ue if ue.stride(2) == 1 else ue.contiguous(), ie if ie.stride(2) == 1 else ie.contiguous(),
                          n_layers, leaky, out_u, out_i, mask_u=mask_u, mask_i=mask_i)
        else:
            scr_u = torch.empty((2, U, d), dtype=torch.float32, device=dev) if n_layers > 1 else None
            scr_i = torch.empty((2, I, d), dtype=torch.float32, device=dev) if n_layers > 1 else None
            for k in range(T):
                ops.gnn_interval(plans_user[k], plans_item[k], ue[k], ie[k], n_layers, leaky, out_u[k], out_i[k], scr_u, scr_i,
                                 mask_u=mask_u[k], mask_i=mask_i[k])
        ctx.save_for_backward(mask_u, mask_i)
        ctx.plans = (plans_user, plans_item)
        ctx.cfg = (n_layers, leaky)
        return out_u, out_i

    @staticmethod
    def backward(ctx, g_user, g_item):
        mask_u, mask_i = ctx.saved_tensors
        plans_user, plans_item = ctx.plans
        n_layers, leaky = ctx.cfg
        T, _, U, dq = mask_u.shape
        I, d, dev = mask_i.shape[2], dq * 4, mask_u.device
        if g_user is None:
            g_user = torch.zeros((T, U, d), dtype=torch.float32, device=dev)
        if g_item is None:
            g_item = torch.zeros((T, I, d), dtype=torch.float32, device=dev)
        if g_user.stride(2) != 1:
            g_user = g_user.contiguous()
        if g_item.stride(2) != 1:
            g_item = g_item.contiguous()
        du = torch.empty((T, U, d), dtype=torch.float32, device=dev)
        di = torch.empty((T, I, d), dtype=torch.float32, device=dev)
        if isinstance(plans_user, ops.SpmmBatch):
            ops.gnn_stack_bwd(plans_user, g_user, g_item, n_layers, leaky, mask_u, mask_i, du, di)
            return du, di, None, None, None, None
        scr_u = torch.empty((4, U, d), dtype=torch.float32, device=dev)
        scr_i = torch.empty((4, I, d), dtype=torch.float32, device=dev)
        for k in range(T):
            ops.gnn_interval_bwd(plans_user[k], plans_item[k], g_user[k], g_item[k], n_layers, leaky, mask_u[k], mask_i[k],
                                 grad_u0=du[k], grad_i0=di[k], scratch_u=scr_u, scratch_i=scr_i)
        return du, di, None, None, None, None


def gnn_stack(u_embed, i_embed, plans_user, plans_item, n_layers: int, leaky: float):
    """plans_user: a list of T ops.SpmmPlan (with plans_item the matching list) or an ops.SpmmBatch (plans_item None)."""
    return GnnStackFn.apply(u_embed, i_embed, plans_user, plans_item, n_layers, leaky)


def _split_qkv_grads(dWqkv, dbqkv, d):
    """[d, 3d] / [3d] -> (dWq, dbq, dWk, dbk, dWv, dbv)."""
    out = ()
    for i in range(3):
        out += (dWqkv[:, i * d:(i + 1) * d].contiguous(), dbqkv[i * d:(i + 1) * d].contiguous())
    return out


def _attn_bwd_front(x, gamma, beta, Wq, bq, Wk, bk, Wv, bv, heads, g_out, want_y=True):
    """sagnn_attn_bwd_front_f32: x [n, t, d] dense, g_out [n, d] -> (y [n*t, d] or None, dqkv [n*t, 3d]).
    gamma None = no layer norm (y = x, not written)."""
    lib = ops._lib.load()
    n, t, d, ld_n, ld_t = ops._ntd("x", x)
    dev = x.device
    dqkv = torch.empty((n * t, 3 * d), dtype=torch.float32, device=dev)
    y = torch.empty((n * t, d), dtype=torch.float32, device=dev) if (want_y and gamma is not None) else None
    vec = lambda name, v, cnt: ops._vec(name, v.detach(), cnt)   # noqa: E731
    ops.check(lib.sagnn_attn_bwd_front_f32(
        x.data_ptr(), ld_n, ld_t, n, t, d, int(heads), None if gamma is None else vec("gamma", gamma, d),
        None if beta is None else vec("beta", beta, d), 1e-12, 0 if gamma is None else 1, vec("Wq", Wq, d * d),
        vec("bq", bq, d), vec("Wk", Wk, d * d), vec("bk", bk, d), vec("Wv", Wv, d * d), vec("bv", bv, d),
        g_out.data_ptr(), int(g_out.stride(0)), dqkv.data_ptr(), ops._ptr(y), ops._stream()))
    return y, dqkv


def lstm_bwd(x, h, gates, cell, dh, drop, W):
    """Whole BPTT in one launch (sagnn_lstm_bwd_f32, d in {32, 64}): x [n, t, d] (any strides),
    h / gates / cell as the training forward stored them, dh [n, t, d] dense = gradient at the
    emitted h -> (dx [n, t, d], dW [2d, 4d], db [4d])."""
    lib = ops._lib.load()
    n, t, d, ld_n, ld_t = ops._ntd("x", x)
    dev = x.device
    dx = torch.empty((n, t, d), dtype=torch.float32, device=dev)
    dW = torch.zeros((2 * d, 4 * d), dtype=torch.float32, device=dev)
    db = torch.zeros(4 * d, dtype=torch.float32, device=dev)
    # scratch for the gate gradients [t, n, 4d]: with it the weight gradient is a second pass on the f16 x 2 engine
    # (sagnn_lstm_bwd_ws_f32) instead of the BPTT launch's fp32-MFMA product
    need = int(lib.sagnn_lstm_bwd_workspace_bytes(n, t, d))
    ws = torch.empty(need // 4, dtype=torch.float32, device=dev) if (SPLIT_DW and t >= SPLIT_DW_MIN_T and 0 < need <= DEFERRED_DW_LIMIT) else None
    ops.check(lib.sagnn_lstm_bwd_ws_f32(x.data_ptr(), ld_n, ld_t, h.data_ptr(), gates.data_ptr(), cell.data_ptr(),
                                        dh.data_ptr(), t * d, ops._ptr(drop), ops._vec("lstm_W", W, 8 * d * d),
                                        dx.data_ptr(), dW.data_ptr(), db.data_ptr(), n, t, d, ops._ptr(ws),
                                        need if ws is not None else 0, ops._stream()))
    return dx, dW, db


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
        out = ops.ln_mhsa_mean(h_emit, ln_gamma.detach(), ln_beta.detach(), Wq.detach(), bq.detach(), Wk.detach(),
                               bk.detach(), Wv.detach(), bv.detach(), heads)
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
        # ---- recompute y and Q|K|V, attention backward -> dQ|dK|dV -----------------------------
        h_emit = h if drop is None else ops.mul(h, drop.contiguous())
        Wqkv = torch.cat([Wq, Wk, Wv], dim=1).detach().contiguous()                      # [d, 3d]
        if lib.sagnn_attn_bwd_front_supported(d, t, heads) and FUSED_ATTN_BWD:
            y2, qkv = _attn_bwd_front(h_emit, ln_gamma.detach(), ln_beta.detach(), Wq, bq, Wk, bk, Wv, bv, heads, g_out)
        else:
            y = ops.layernorm_td(h_emit, ln_gamma.detach(), ln_beta.detach())            # [n, t, d]
            bqkv = torch.cat([bq, bk, bv]).detach().contiguous()
            y2 = y.view(n * t, d)
            qkv = ops.dense_nn(y2, Wqkv, bqkv)                                           # [n*t, 3d]
            ops.check(lib.sagnn_attn_bwd_f32(qkv.data_ptr(), g_out.data_ptr(), d, n, t, d, heads, st))
        dWqkv = torch.zeros((d, 3 * d), dtype=torch.float32, device=dev)
        dbqkv = torch.zeros(3 * d, dtype=torch.float32, device=dev)
        if lib.sagnn_attn_bwd_tail_supported(d) and FUSED_ATTN_BWD:
            # dW += y^T dQKV, db += colsum dQKV and dy = dQKV W^T (over y) in one pass over dQKV
            ops.check(lib.sagnn_attn_bwd_tail_f32(y2.data_ptr(), qkv.data_ptr(), n * t, d, Wqkv.data_ptr(),
                                                  dWqkv.data_ptr(), dbqkv.data_ptr(), st))
            dy = y2
        else:
            ops.dense_tn(y2, qkv, dWqkv, dbqkv)
            dy = ops.dense_nn(qkv, Wqkv.t().contiguous(), None, out=y2)                  # reuses y's storage
        # ---- layer norm backward (in place on dy) ----------------------------------------------
        dgamma = torch.zeros(d, dtype=torch.float32, device=dev)
        dbeta = torch.zeros(d, dtype=torch.float32, device=dev)
        dh = dy.view(n, t, d)
        ops.check(lib.sagnn_layernorm_td_bwd_f32(h_emit.data_ptr(), t * d, dh.data_ptr(), t * d, n, t, d,
                                                 ops._vec("gamma", ln_gamma.detach(), d), 1e-12, dh.data_ptr(),
                                                 t * d, dgamma.data_ptr(), dbeta.data_ptr(), st))
        # ---- BPTT ----------------------------------------------------------------------------------
        if lib.sagnn_lstm_bwd_supported(d) and FUSED_BPTT:
            dx, dW, db = lstm_bwd(x, h, gates, cell, dh, drop, lstm_W.detach())
            return (dx, dW, db, dgamma, dbeta) + _split_qkv_grads(dWqkv, dbqkv, d) + (None, None)
        # generic BPTT (any d that is a multiple of 32; d = 128 is BASELINE config 3). Per step: the element-wise gate
        # backward and ONE product d[x_t | h_{t-1}] = dG_t W^T written where the next step reads it. The gate gradients of
        # all steps stay in HBM ([t, n, 4d]) and the weight gradient is two segmented products after the loop instead of
        # 2 t small ones (each of those a split-K launch ending in 64 K float atomics per block).
        WT = lstm_W.detach().t().contiguous()                                            # [4d, 2d]
        dW = torch.zeros((2 * d, 4 * d), dtype=torch.float32, device=dev)
        db = torch.zeros(4 * d, dtype=torch.float32, device=dev)
        defer = n * t * 4 * d * 4 <= DEFERRED_DW_LIMIT
        dG = torch.empty((t if defer else 1, n, 4 * d), dtype=torch.float32, device=dev)
        dc = [torch.empty((n, d), dtype=torch.float32, device=dev) for _ in range(2)]
        dxh = torch.empty((n, t, 2 * d), dtype=torch.float32, device=dev)                # [dx_t | dh_{t-1}] per step
        for ts in range(t - 1, -1, -1):
            last = ts == t - 1
            dgates = dG[ts if defer else 0]
            ops.check(lib.sagnn_lstm_bwd_step_f32(
                gates.data_ptr(), cell.data_ptr(), dh.data_ptr(), t * d, ops._ptr(drop),
                None if last else dxh[:, ts + 1, d:].data_ptr(), t * 2 * d, None if last else dc[(ts + 1) & 1].data_ptr(),
                dgates.data_ptr(), dc[ts & 1].data_ptr(), n, t, d, ts, st))
            if not defer:
                ops.dense_tn(x[:, ts, :], dgates, dW[:d], db)
                if ts > 0:
                    ops.dense_tn(h[:, ts - 1, :], dgates, dW[d:], None)                  # h un-dropped: the recurrent operand
            ops.dense_nn(dgates, WT, None, out=dxh[:, ts, :])
        if defer:
            ops.dense_tn_seg(x.permute(1, 0, 2), dG, dW[:d], db)
            if t > 1:
                ops.dense_tn_seg(h.permute(1, 0, 2)[:t - 1], dG[1:], dW[d:], None)
        return (dxh[:, :, :d].contiguous(), dW, db, dgamma, dbeta) + _split_qkv_grads(dWqkv, dbqkv, d) + (None, None)


def interval_fusion(x, p: dict, heads: int, drop_scale=None):
    """Differentiable interval fusion; p as in ops.interval_fusion."""
    return IntervalFusionFn.apply(x, p["lstm_W"], p["lstm_b"], p["ln_gamma"], p["ln_beta"], p["Wq"], p["bq"],
                                  p["Wk"], p["bk"], p["Wv"], p["bv"], heads, drop_scale)


# ----------------------------------------------------------------------------------------------
# Stand-alone differentiable pieces (prediction head, SSL branch): model.py:156-205, 241-250
# ----------------------------------------------------------------------------------------------


def _mhsa_mean_backward(y, Wq, bq, Wk, bk, Wv, bv, heads, g_out):
    """y [n, t, d] dense, g_out [n, d] -> (dy [n, t, d], dWq, dbq, dWk, dbk, dWv, dbv)."""
    lib = ops._lib.load()
    n, t, d = y.shape
    dev = y.device
    Wqkv = torch.cat([Wq, Wk, Wv], dim=1).detach().contiguous()
    bqkv = torch.cat([bq, bk, bv]).detach().contiguous()
    y2 = y.reshape(n * t, d)
    if lib.sagnn_attn_bwd_front_supported(d, t, heads) and FUSED_ATTN_BWD:
        _, qkv = _attn_bwd_front(y.contiguous(), None, None, Wq, bq, Wk, bk, Wv, bv, heads, g_out.contiguous())
    else:
        qkv = ops.dense_nn(y2, Wqkv, bqkv)
        ops.check(lib.sagnn_attn_bwd_f32(qkv.data_ptr(), g_out.data_ptr(), d, n, t, d, heads, ops._stream()))
    dWqkv = torch.zeros((d, 3 * d), dtype=torch.float32, device=dev)
    dbqkv = torch.zeros(3 * d, dtype=torch.float32, device=dev)
    ops.dense_tn(y2, qkv, dWqkv, dbqkv)
    dy = ops.dense_nn(qkv, Wqkv.t().contiguous(), None).view(n, t, d)
    dWs = [dWqkv[:, i * d:(i + 1) * d].contiguous() for i in range(3)]
    dbs = [dbqkv[i * d:(i + 1) * d].contiguous() for i in range(3)]
    return dy, dWs[0], dbs[0], dWs[1], dbs[1], dWs[2], dbs[2]


class SpmmFn(torch.autograd.Function):
    """y = A·x (pattern sum, no activation): the masked sums of model.py:161-162 on a per-batch
    CSR. Backward is the same kernel on the transposed CSR."""

    @staticmethod
    def forward(ctx, x, plan, plan_t):
        ctx.plan_t = plan_t
        return ops.spmm(plan, x.detach().contiguous(), 1.0)

    @staticmethod
    def backward(ctx, g):
        return ops.spmm(ctx.plan_t, g.contiguous(), 1.0), None, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        x = x.detach().contiguous()
        ctx.save_for_backward(x, gamma)
        return ops.layernorm_td(x, gamma.detach(), beta.detach())

    @staticmethod
    def backward(ctx, g):
        x, gamma = ctx.saved_tensors
        n, t, d = x.shape
        g = g.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.zeros(d, dtype=torch.float32, device=x.device)
        dbeta = torch.zeros(d, dtype=torch.float32, device=x.device)
        ops.check(ops._lib.load().sagnn_layernorm_td_bwd_f32(
            x.data_ptr(), t * d, g.data_ptr(), t * d, n, t, d, ops._vec("gamma", gamma.detach(), d), 1e-12,
            dx.data_ptr(), t * d, dgamma.data_ptr(), dbeta.data_ptr(), ops._stream()))
        return dx, dgamma, dbeta


class MhsaMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Wq, bq, Wk, bk, Wv, bv, heads):
        x = x.detach().contiguous()
        ctx.save_for_backward(x, Wq, bq, Wk, bk, Wv, bv)
        ctx.heads = heads
        return ops.mhsa_mean(x, Wq.detach(), bq.detach(), Wk.detach(), bk.detach(), Wv.detach(), bv.detach(), heads)

    @staticmethod
    def backward(ctx, g):
        x, Wq, bq, Wk, bk, Wv, bv = ctx.saved_tensors
        return _mhsa_mean_backward(x, Wq, bq, Wk, bk, Wv, bv, ctx.heads, g.contiguous()) + (None,)


class LeakyAddFn(torch.autograd.Function):
    """out = max(leaky*a, a) + b (model.py:166)."""

    @staticmethod
    def forward(ctx, a, b, leaky):
        a = a.detach().contiguous()
        ctx.save_for_backward(a)
        ctx.leaky = leaky
        return ops.leaky_add(a, b.detach().contiguous(), leaky)

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        g = g.contiguous()
        da = torch.empty_like(a)
        ops.check(ops._lib.load().sagnn_leaky_f32(a.data_ptr(), g.data_ptr(), da.data_ptr(), ctx.leaky, a.numel(), 1,
                                                  ops._stream()))
        return da, g, None


class PairScoreFn(torch.autograd.Function):
    """preds[e] = <U[u], I[i]> + <leaky(S[l]), I[i]> (model.py:169-173; iEmbed_att IS final_item_vector)."""

    @staticmethod
    def forward(ctx, U, I, S, uids, iids, locs, leaky):
        U, I, S = U.detach().contiguous(), I.detach().contiguous(), S.detach().contiguous()
        ctx.save_for_backward(U, I, S, uids, iids, locs)
        ctx.leaky = leaky
        return ops.pair_score(U, I, uids, iids, S=S, A=I, locs=locs, leaky=leaky)

    @staticmethod
    def backward(ctx, g):
        U, I, S, uids, iids, locs = ctx.saved_tensors
        d = U.shape[1]
        dU, dI, dS = torch.zeros_like(U), torch.zeros_like(I), torch.zeros_like(S)
        g = g.contiguous()
        ops.check(ops._lib.load().sagnn_pair_score_bwd_f32(
            U.data_ptr(), d, I.data_ptr(), d, S.data_ptr(), d, I.data_ptr(), d, uids.data_ptr(), iids.data_ptr(),
            locs.data_ptr(), ctx.leaky, g.data_ptr(), dU.data_ptr(), dI.data_ptr(), dS.data_ptr(), dI.data_ptr(),
            uids.numel(), d, ops._stream()))
        return dU, dI, dS, None, None, None, None


class ProdLeakySumFn(torch.autograd.Function):
    """s[e] = sum_j leaky(X[u][j] * Y[i][j]) (model.py:191, :199)."""

    @staticmethod
    def forward(ctx, X, Y, uids, iids, leaky):
        X, Y = X.detach().contiguous(), Y.detach().contiguous()
        ctx.save_for_backward(X, Y, uids, iids)
        ctx.leaky = leaky
        d = X.shape[1]
        out = torch.empty(uids.numel(), dtype=torch.float32, device=X.device)
        ops.check(ops._lib.load().sagnn_prod_leaky_sum_f32(X.data_ptr(), d, Y.data_ptr(), d, uids.data_ptr(),
                                                           iids.data_ptr(), leaky, out.data_ptr(), uids.numel(), d,
                                                           ops._stream()))
        return out

    @staticmethod
    def backward(ctx, g):
        X, Y, uids, iids = ctx.saved_tensors
        d = X.shape[1]
        dX, dY = torch.zeros_like(X), torch.zeros_like(Y)
        g = g.contiguous()
        ops.check(ops._lib.load().sagnn_prod_leaky_sum_bwd_f32(X.data_ptr(), d, Y.data_ptr(), d, uids.data_ptr(),
                                                               iids.data_ptr(), ctx.leaky, g.data_ptr(), dX.data_ptr(),
                                                               dY.data_ptr(), uids.numel(), d, ops._stream()))
        return dX, dY, None, None, None


def _pad_cols(W, kp):
    out = torch.zeros(W.shape[:-1] + (kp,), dtype=W.dtype, device=W.device)
    out[..., : W.shape[-1]] = W
    return out


class MetaWeightFn(torch.autograd.Function):
    """w[e] = sigmoid(FC(leaky(FC([F*V | F | V][u_e])))) (model.py:179-182), evaluated for the sampled
    users only (the reference computes it for every user and gathers afterwards: same values).
    W2 [3d, k], b2 [k], W3 [k, 1], b3 [1]; k = ssldim is padded to a multiple of 32 for the MFMA
    product."""

    @staticmethod
    def forward(ctx, F, V, uids, W2, b2, W3, b3, leaky):
        lib = ops._lib.load()
        F, V = F.detach().contiguous(), V.detach().contiguous()
        n, d, k = uids.numel(), F.shape[1], W2.shape[1]
        kp = (k + 31) // 32 * 32
        dev = F.device
        m1 = torch.empty((n, 3 * d), dtype=torch.float32, device=dev)
        ops.check(lib.sagnn_meta_features_f32(F.data_ptr(), d, V.data_ptr(), d, uids.data_ptr(), m1.data_ptr(), n, d,
                                              ops._stream()))
        W2p = _pad_cols(W2.detach(), kp).contiguous()
        z1 = ops.dense_nn(m1, W2p, _pad_cols(b2.detach(), kp).contiguous())
        a1 = torch.empty_like(z1)
        ops.check(lib.sagnn_leaky_f32(z1.data_ptr(), None, a1.data_ptr(), leaky, z1.numel(), 0, ops._stream()))
        w = torch.empty(n, dtype=torch.float32, device=dev)
        w3 = W3.detach().reshape(-1).contiguous()
        ops.check(lib.sagnn_rowdot_sigmoid_f32(a1.data_ptr(), kp, w3.data_ptr(), b3.detach().data_ptr(), w.data_ptr(),
                                               n, k, ops._stream()))
        ctx.save_for_backward(F, V, uids, m1, z1, a1, w, W2p, w3)
        ctx.cfg = (leaky, k, kp)
        return w

    @staticmethod
    def backward(ctx, dw):
        lib = ops._lib.load()
        F, V, uids, m1, z1, a1, w, W2p, w3 = ctx.saved_tensors
        leaky, k, kp = ctx.cfg
        n, d = uids.numel(), F.shape[1]
        dev = F.device
        dw = dw.contiguous()
        dA = torch.zeros_like(a1)
        dw3 = torch.zeros(k, dtype=torch.float32, device=dev)
        db3 = torch.zeros(1, dtype=torch.float32, device=dev)
        ops.check(lib.sagnn_rowdot_sigmoid_bwd_f32(a1.data_ptr(), kp, w3.data_ptr(), w.data_ptr(), dw.data_ptr(),
                                                   dA.data_ptr(), kp, dw3.data_ptr(), db3.data_ptr(), n, k,
                                                   ops._stream()))
        dz1 = torch.empty_like(z1)
        ops.check(lib.sagnn_leaky_f32(z1.data_ptr(), dA.data_ptr(), dz1.data_ptr(), leaky, z1.numel(), 1, ops._stream()))
        dW2p = torch.zeros((3 * d, kp), dtype=torch.float32, device=dev)
        db2p = torch.zeros(kp, dtype=torch.float32, device=dev)
        ops.dense_tn(m1, dz1, dW2p, db2p)
        dm1 = ops.dense_nn(dz1, W2p.t().contiguous(), None)
        dF, dV = torch.zeros_like(F), torch.zeros_like(V)
        ops.check(lib.sagnn_meta_features_bwd_f32(F.data_ptr(), d, V.data_ptr(), d, uids.data_ptr(), dm1.data_ptr(),
                                                  dF.data_ptr(), dV.data_ptr(), n, d, ops._stream()))
        return dF, dV, None, dW2p[:, :k].contiguous(), db2p[:k].contiguous(), dw3.view(k, 1), db3, None


class HingeFn(torch.autograd.Function):
    """scale * sum max(0, 1 - S*(pos - neg)), S = wp*sp - wn*sn or 1 (model.py:202, :244). sp/sn are
    constants (tf.stop_gradient, model.py:192-193)."""

    @staticmethod
    def forward(ctx, pos, neg, wp, wn, sp, sn, scale):
        lib = ops._lib.load()
        pos, neg = pos.detach().contiguous(), neg.detach().contiguous()
        n = pos.numel()
        dev = pos.device
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        dpos, dneg = torch.empty_like(pos), torch.empty_like(neg)
        weighted = wp is not None
        if weighted:
            wp, wn = wp.detach().contiguous(), wn.detach().contiguous()
            sp, sn = sp.detach().contiguous(), sn.detach().contiguous()
            dwp, dwn = torch.empty_like(wp), torch.empty_like(wn)
        ops.check(lib.sagnn_hinge_f32(pos.data_ptr(), neg.data_ptr(), ops._ptr(wp) if weighted else None,
                                      ops._ptr(wn) if weighted else None, ops._ptr(sp) if weighted else None,
                                      ops._ptr(sn) if weighted else None, float(scale), loss.data_ptr(),
                                      dpos.data_ptr(), dneg.data_ptr(), dwp.data_ptr() if weighted else None,
                                      dwn.data_ptr() if weighted else None, n, ops._stream()))
        ctx.weighted = weighted
        ctx.save_for_backward(dpos, dneg, *((dwp, dwn) if weighted else ()))
        return loss

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        dpos, dneg = saved[0] * g, saved[1] * g
        if ctx.weighted:
            return dpos, dneg, saved[2] * g, saved[3] * g, None, None, None
        return dpos, dneg, None, None, None, None, None
