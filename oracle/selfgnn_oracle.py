"""CPU restatement of SelfGNN's interval-propagation hot path — TEST INFRASTRUCTURE ONLY.

This module is the checker for the HIP kernels in sa-gnn_amd/csrc. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product path never
does (sa-gnn_amd/ raises if libsagnn.so is missing instead of falling back here).

Every function restates one piece of the reference (LIU-YUXI/SA-GNN, TF1) op for op in numpy,
fp32 unless `dtype` says otherwise, and cites the file:line it follows. The arithmetic of the
reference lives in TensorFlow 1.14.0 (requirements.txt:4), which is not installable here
(Python 3.10, no network), so TF-op semantics are restated from their documented behaviour:
GatherV2, SegmentSum (output rows = max(id)+1, sequential sum per segment), Pad, Maximum, AddN,
BasicLSTMCell (gate order i, j, f, o; forget_bias 1.0), contrib.layers.layer_norm
(begin_norm_axis=1, begin_params_axis=-1, eps 1e-12), layers.dense.

PARITY PINNING: the reference ships no tests, golden vectors or expected outputs (SURVEY.md §4,
§8c). What IS pinned against the running reference: transToLsts / transpose (DataHandler.py
imports here; tests/golden/make_golden.py records their outputs). The TF-side arithmetic is
"parity unpinned": it is cross-checked against an independent formulation (scipy pattern-CSR @
dense, torch.nn.LSTMCell-free hand-rolled recurrences) in tests/test_oracle.py, not against TF.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

# ----------------------------------------------------------------------------------------------
# Data side: DataHandler.py:9-11, 47-69
# ----------------------------------------------------------------------------------------------


def transpose(mat):
    """DataHandler.transpose (DataHandler.py:9-11): CSR -> COO -> transpose -> CSR.

    scipy's COO->CSR conversion sums duplicate (row, col) entries and sorts each row, so a
    duplicated edge counts once in the transposed adjacency but twice in the forward one."""
    return sp.csr_matrix(sp.coo_matrix(mat).transpose())


def trans_to_lsts(mat, norm: bool = False):
    """DataHandler.transToLsts (DataHandler.py:47-69), mask=False.

    Returns (indices int32 [nnz, 2] as (row, col) in stored CSR order, data int32 [nnz], shape).
    With norm=True the reference multiplies the degree normalisers into an int32 array
    (DataHandler.py:51, 56-59), so every value truncates toward zero; the values are dead
    anyway (model.py:84). An empty matrix yields one phantom edge (0, 0) (DataHandler.py:66-68)."""
    shape = [mat.shape[0], mat.shape[1]]
    coo = sp.coo_matrix(mat)
    indices = np.stack([coo.row, coo.col], axis=1).astype(np.int32).reshape(-1, 2)
    data = coo.data.astype(np.int32)
    if norm and len(data):
        row_d = np.squeeze(np.array(1 / (np.sqrt(np.sum(mat, axis=1) + 1e-8) + 1e-8)), axis=1)
        col_d = np.squeeze(np.array(1 / (np.sqrt(np.sum(mat, axis=0) + 1e-8) + 1e-8)), axis=0)
        # element-wise, in the reference's order: int32 <- int32 * float64 * float64 (truncates)
        scaled = data.astype(np.float64) * row_d[indices[:, 0]] * col_d[indices[:, 1]]
        data = np.trunc(scaled).astype(np.int32)
    if indices.shape[0] == 0:
        indices = np.array([[0, 0]], dtype=np.int32)
        data = np.array([0], dtype=np.int32)
    return indices, data, shape


# ----------------------------------------------------------------------------------------------
# Propagation: model.py:80-92, 118-134
# ----------------------------------------------------------------------------------------------


def segment_sum(data: np.ndarray, segment_ids: np.ndarray) -> np.ndarray:
    """tf.math.segment_sum: sorted ids, output has max(id)+1 rows, missing ids give zero rows.
    Each segment is accumulated sequentially in input order (the TF CPU kernel's order)."""
    if len(segment_ids) == 0:
        return np.zeros((0, data.shape[1]), dtype=data.dtype)
    if np.any(np.diff(segment_ids) < 0):
        raise ValueError("segment ids are not sorted")  # TF raises InvalidArgument
    n_out = int(segment_ids[-1]) + 1
    out = np.zeros((n_out, data.shape[1]), dtype=data.dtype)
    for e in range(len(segment_ids)):  # pure-Python loop: small cases only
        out[segment_ids[e]] += data[e]
    return out


def segment_sum_fast(data: np.ndarray, segment_ids: np.ndarray) -> np.ndarray:
    """Same result up to fp32 summation order (np.add.reduceat); for the larger test cases."""
    if len(segment_ids) == 0:
        return np.zeros((0, data.shape[1]), dtype=data.dtype)
    n_out = int(segment_ids[-1]) + 1
    out = np.zeros((n_out, data.shape[1]), dtype=data.dtype)
    starts = np.flatnonzero(np.r_[True, np.diff(segment_ids) != 0])
    out[segment_ids[starts]] = np.add.reduceat(data, starts, axis=0)
    return out


def leaky_relu(x: np.ndarray, leaky: float) -> np.ndarray:
    """ActivateHelp('leakyRelu') = tf.maximum(leaky*data, data) (Utils/NNLayers.py:135-136)."""
    return np.maximum(np.asarray(leaky, dtype=x.dtype) * x, x)


def message_propagate(srclats: np.ndarray, indices: np.ndarray, n_out: int, leaky: float,
                      exact_order: bool = False) -> np.ndarray:
    """Recommender.messagePropagate (model.py:80-92).

    src = indices[:, 1], tgt = indices[:, 0] (:82-83); gather (:86); segment_sum + 100 rows of
    zero padding (:87); rows 0..n_out-1 (:88-91); leaky-ReLU (:92). TF-CPU raises when n_out
    exceeds max(tgt)+1+100 (GatherV2 out of range); so does this."""
    src = indices[:, 1]
    tgt = indices[:, 0]
    gathered = srclats[src]                                            # GatherV2 [nnz, d]
    seg = (segment_sum if exact_order else segment_sum_fast)(gathered, tgt)
    lat = np.concatenate([seg, np.zeros((100, srclats.shape[1]), dtype=srclats.dtype)], axis=0)
    if n_out > lat.shape[0]:
        raise IndexError(
            f"gather of rows 0..{n_out - 1} from {lat.shape[0]} rows: the last connected row is "
            "more than 100 below N (TF-CPU InvalidArgument; TF-GPU returns zeros)")
    return leaky_relu(lat[:n_out], leaky)


def message_propagate_zero_fill(srclats, indices, n_out, leaky):
    """The build's contract for the >100-trailing-empty-rows case: exactly
    [n_out, d], isolated rows are leaky(0) = 0 (what TF-GPU returns)."""
    src = indices[:, 1]
    tgt = indices[:, 0]
    seg = segment_sum_fast(srclats[src], tgt)
    lat = np.zeros((n_out, srclats.shape[1]), dtype=srclats.dtype)
    m = min(n_out, seg.shape[0])
    lat[:m] = seg[:m]
    return leaky_relu(lat, leaky)


def gnn_interval(u0, i0, adj_idx, tp_idx, n_layers: int, leaky: float, zero_fill: bool = True):
    """One k of the loop model.py:118-129: both directions read layer l (simultaneous update,
    a_emb1 is built from embs0[-1] before :124 appends), residual adds (:124-125), add_n (:126-127)."""
    mp = message_propagate_zero_fill if zero_fill else message_propagate
    embs0, embs1 = [u0], [i0]
    for _ in range(n_layers):
        a0 = mp(embs1[-1], adj_idx, u0.shape[0], leaky)
        a1 = mp(embs0[-1], tp_idx, i0.shape[0], leaky)
        embs0.append(a0 + embs0[-1])
        embs1.append(a1 + embs1[-1])
    user = embs0[0]
    for e in embs0[1:]:
        user = user + e                                                # AddN, left to right
    item = embs1[0]
    for e in embs1[1:]:
        item = item + e
    return user, item


def gnn_stack(u_embed, i_embed, adj_list, tp_list, n_layers: int, leaky: float):
    """model.py:118-134: all T intervals, tf.stack + tf.transpose -> [N, T, d]."""
    users, items = [], []
    for k in range(len(adj_list)):
        u, i = gnn_interval(u_embed[k], i_embed[k], adj_list[k], tp_list[k], n_layers, leaky)
        users.append(u)
        items.append(i)
    return np.stack(users, 0).transpose(1, 0, 2), np.stack(items, 0).transpose(1, 0, 2)


# ----------------------------------------------------------------------------------------------
# Interval fusion: model.py:135-155, Utils/attention.py:31-78
# ----------------------------------------------------------------------------------------------


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def basic_lstm(x: np.ndarray, kernel: np.ndarray, bias: np.ndarray, forget_bias: float = 1.0,
               drop_scale: np.ndarray | None = None) -> np.ndarray:
    """dynamic_rnn over MultiRNNCell([DropoutWrapper(BasicLSTMCell(d))]) (model.py:135-146).

    TF 1.14 BasicLSTMCell.call: gate_inputs = concat([x_t, h], 1) @ kernel + bias;
    i, j, f, o = split(gate_inputs, 4, axis=1);
    new_c = c*sigmoid(f + forget_bias) + sigmoid(i)*tanh(j); new_h = tanh(new_c)*sigmoid(o).
    Zero initial state. DropoutWrapper(output_keep_prob) scales only the emitted h."""
    n, t, d = x.shape
    h = np.zeros((n, d), dtype=x.dtype)
    c = np.zeros((n, d), dtype=x.dtype)
    fb = np.asarray(forget_bias, dtype=x.dtype)
    outs = []
    for ts in range(t):
        g = np.concatenate([x[:, ts, :], h], axis=1) @ kernel + bias
        gi, gj, gf, go = np.split(g, 4, axis=1)
        c = c * _sigmoid(gf + fb) + _sigmoid(gi) * np.tanh(gj)
        h = np.tanh(c) * _sigmoid(go)
        outs.append(h if drop_scale is None else h * drop_scale[:, ts, :])
    return np.stack(outs, axis=1)


def layer_norm_td(x: np.ndarray, gamma: np.ndarray, beta: np.ndarray, eps: float = 1e-12):
    """tf.contrib.layers.layer_norm defaults (model.py:152-153): moments over axes (1, 2) per
    node (tf.nn.moments: population variance around the mean), then tf.nn.batch_normalization:
    inv = rsqrt(var + eps) * gamma;  y = x*inv + (beta - mean*inv)."""
    mean = x.mean(axis=(1, 2), keepdims=True, dtype=x.dtype)
    var = np.square(x - mean).mean(axis=(1, 2), keepdims=True, dtype=x.dtype)
    inv = (1.0 / np.sqrt(var + np.asarray(eps, dtype=x.dtype))).astype(x.dtype) * gamma
    return x * inv + (beta - mean * inv)


def mhsa(x, wq, bq, wk, bk, wv, bv, heads: int):
    """MultiHeadSelfAttention.attention (Utils/attention.py:55-78) with
    ScaledDotProductAttention.attention (:35-45): three tf.layers.dense with bias, reshape to
    [N, heads, T, d_k], scores = exp(QK^T/sqrt(d_k)) with no max subtraction (:38-39),
    attn = scores/(sum + 1e-8) (:43), context = attn @ V (:44), heads merged (:77)."""
    n, t, d = x.shape
    dk = d // heads
    q = (x @ wq + bq).reshape(n, t, heads, dk).transpose(0, 2, 1, 3)
    k = (x @ wk + bk).reshape(n, t, heads, dk).transpose(0, 2, 1, 3)
    v = (x @ wv + bv).reshape(n, t, heads, dk).transpose(0, 2, 1, 3)
    scores = np.exp((q @ k.transpose(0, 1, 3, 2)) / np.asarray(np.sqrt(dk), dtype=x.dtype))
    attn = scores / (scores.sum(axis=-1, keepdims=True) + np.asarray(1e-8, dtype=x.dtype))
    ctx = attn @ v
    return ctx.transpose(0, 2, 1, 3).reshape(n, t, heads * dk)


def interval_fusion(x, p: dict, heads: int):
    """model.py:135-155 for one node type: LSTM -> layer_norm -> MHSA -> reduce_mean(axis=1).
    p holds lstm_W [2d,4d], lstm_b [4d], ln_gamma [d], ln_beta [d], Wq/bq/Wk/bk/Wv/bv."""
    h = basic_lstm(x, p["lstm_W"], p["lstm_b"], 1.0)
    y = layer_norm_td(h, p["ln_gamma"], p["ln_beta"], 1e-12)
    a = mhsa(y, p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], heads)
    return a.mean(axis=1, dtype=x.dtype)


# ----------------------------------------------------------------------------------------------
# Parameter initialisers: Utils/NNLayers.py:43-68 (xavier), TF defaults for LSTM / dense / LN
# ----------------------------------------------------------------------------------------------


def xavier_uniform(shape, rng: np.random.Generator, dtype=np.float32):
    """tf.contrib.layers.xavier_initializer(uniform=True): limit = sqrt(6/(fan_in+fan_out));
    for rank > 2 both fans are multiplied by prod(shape[:-2]) (so [T, N, d] has fan_in = T*N)."""
    shape = tuple(int(s) for s in shape)
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def init_fusion_params(d: int, rng: np.random.Generator, dtype=np.float32) -> dict:
    """Shapes and initial values as TF creates them: BasicLSTMCell kernel [2d, 4d] glorot
    uniform, bias zeros; layer_norm gamma ones / beta zeros; dense kernels xavier, biases zeros.
    Biases and beta get small random values instead of zeros so tests exercise them."""
    return {
        "lstm_W": xavier_uniform((2 * d, 4 * d), rng, dtype),
        "lstm_b": (0.1 * rng.standard_normal(4 * d)).astype(dtype),
        "ln_gamma": (1.0 + 0.1 * rng.standard_normal(d)).astype(dtype),
        "ln_beta": (0.1 * rng.standard_normal(d)).astype(dtype),
        "Wq": xavier_uniform((d, d), rng, dtype), "bq": (0.1 * rng.standard_normal(d)).astype(dtype),
        "Wk": xavier_uniform((d, d), rng, dtype), "bk": (0.1 * rng.standard_normal(d)).astype(dtype),
        "Wv": xavier_uniform((d, d), rng, dtype), "bv": (0.1 * rng.standard_normal(d)).astype(dtype),
    }


# ----------------------------------------------------------------------------------------------
# Gradient oracle: the same path restated in differentiable torch (CPU, float64) so that
# torch.autograd supplies what tf.gradients would (reference model.py:250 minimises the loss
# through these ops). Test infrastructure only.
# ----------------------------------------------------------------------------------------------


def torch_message_propagate(srclats, indices, n_out: int, leaky: float):
    """model.py:80-92 in torch: gather, sorted segment sum (index_add), zero-filled to n_out rows,
    tf.maximum(leaky*x, x). On ties tf.maximum sends the gradient to its first argument
    (MaximumGrad uses x >= y), restated with torch.where."""
    import torch
    idx = torch.as_tensor(indices, dtype=torch.long)
    gathered = srclats.index_select(0, idx[:, 1])
    lat = torch.zeros((n_out, srclats.shape[1]), dtype=srclats.dtype).index_add(0, idx[:, 0], gathered)
    a = leaky * lat
    return torch.where(a >= lat, a, lat)


def torch_gnn_interval(u0, i0, adj_idx, tp_idx, n_layers: int, leaky: float):
    """model.py:118-129 in torch (see gnn_interval above)."""
    embs0, embs1 = [u0], [i0]
    for _ in range(n_layers):
        a0 = torch_message_propagate(embs1[-1], adj_idx, u0.shape[0], leaky)
        a1 = torch_message_propagate(embs0[-1], tp_idx, i0.shape[0], leaky)
        embs0.append(a0 + embs0[-1])
        embs1.append(a1 + embs1[-1])
    return sum(embs0[1:], embs0[0]), sum(embs1[1:], embs1[0])


def torch_basic_lstm(x, kernel, bias, forget_bias: float = 1.0):
    """basic_lstm above, in differentiable torch."""
    import torch
    n, t, d = x.shape
    h = torch.zeros((n, d), dtype=x.dtype)
    c = torch.zeros((n, d), dtype=x.dtype)
    outs = []
    for ts in range(t):
        g = torch.cat([x[:, ts, :], h], dim=1) @ kernel + bias
        gi, gj, gf, go = torch.split(g, d, dim=1)
        c = c * torch.sigmoid(gf + forget_bias) + torch.sigmoid(gi) * torch.tanh(gj)
        h = torch.tanh(c) * torch.sigmoid(go)
        outs.append(h)
    return torch.stack(outs, dim=1)


def torch_layer_norm_td(x, gamma, beta, eps: float = 1e-12):
    """layer_norm_td above, in differentiable torch (tf.nn.moments stops the gradient through the
    mean inside the variance; the total derivative is the same)."""
    import torch
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    inv = torch.rsqrt(var + eps) * gamma
    return x * inv + (beta - mean * inv)


def torch_mhsa_mean(x, wq, bq, wk, bk, wv, bv, heads: int):
    """mhsa above followed by reduce_mean(axis=1), in differentiable torch."""
    import torch
    n, t, d = x.shape
    dk = d // heads
    q = (x @ wq + bq).reshape(n, t, heads, dk).permute(0, 2, 1, 3)
    k = (x @ wk + bk).reshape(n, t, heads, dk).permute(0, 2, 1, 3)
    v = (x @ wv + bv).reshape(n, t, heads, dk).permute(0, 2, 1, 3)
    scores = torch.exp((q @ k.transpose(-1, -2)) / float(np.sqrt(dk)))
    attn = scores / (scores.sum(dim=-1, keepdim=True) + 1e-8)
    ctx = (attn @ v).permute(0, 2, 1, 3).reshape(n, t, d)
    return ctx.mean(dim=1)


def torch_interval_fusion(x, p: dict, heads: int):
    """interval_fusion above in differentiable torch; p holds torch tensors."""
    h = torch_basic_lstm(x, p["lstm_W"], p["lstm_b"], 1.0)
    y = torch_layer_norm_td(h, p["ln_gamma"], p["ln_beta"], 1e-12)
    return torch_mhsa_mean(y, p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], heads)


# ----------------------------------------------------------------------------------------------
# Prediction head and metrics: model.py:156-173, 484-510
# ----------------------------------------------------------------------------------------------


def prediction_head(final_user, final_item, pos_embed, ln_params, att_params, uids, iids, sequence, mask,
                    ulocs_seq, heads: int, leaky: float):
    """model.py:156-173. ln_params: list of (gamma, beta) in creation order — [0] item-sequence
    token, [1] position token, [2 + i] attention layer i; att_params: list of dicts Wq..bv.
    The sequence is collapsed to ONE token per batch slot by the masked sums (:161-162), so every
    attention layer runs on length-1 sequences."""
    m = mask[:, None, :].astype(final_item.dtype)                                    # [B, 1, L]
    seq = m @ final_item[sequence]                                                    # [B, 1, d]
    pos = m @ np.broadcast_to(pos_embed[None, :, :], (mask.shape[0],) + pos_embed.shape)
    sb = layer_norm_td(seq, *ln_params[0]) + layer_norm_td(pos, *ln_params[1])
    att = sb
    for i, p in enumerate(att_params):
        a1 = mhsa(layer_norm_td(att, *ln_params[2 + i]), p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], heads)
        att = leaky_relu(a1, leaky) + att
    att_user = att.sum(axis=1)                                                        # [B, d]
    pck_u, pck_i = final_user[uids], final_item[iids]
    preds = (pck_u * pck_i).sum(-1)
    return preds + (leaky_relu(att_user[ulocs_seq], leaky) * pck_i).sum(-1)


def calc_res(preds, tem_tst, tst_locs, shoot: int = 10):
    """Recommender.calcRes (model.py:484-510), the reference's own loop: Python's stable sort with
    reverse=True keeps the original order among equal scores, and the positive is the LAST
    candidate, so it loses ties. Returns (hit@shoot, ndcg@shoot, hit@5, ndcg@5, hit@20, ndcg@20)."""
    out = [0.0] * 6
    for j in range(preds.shape[0]):
        predvals = list(zip(preds[j], tst_locs[j]))
        predvals.sort(key=lambda x: x[0], reverse=True)
        for slot, k in enumerate((shoot, 5, 20)):
            top = [x[1] for x in predvals[:k]]
            if tem_tst[j] in top:
                out[2 * slot] += 1
                out[2 * slot + 1] += float(np.reciprocal(np.log2(top.index(tem_tst[j]) + 2)))
    return tuple(out)


def torch_train_loss(P: dict, adj_list, tp_list, batch: dict, cfg: dict):
    """The reference's training objective for one step (model.py:104-205, 241-246) in differentiable
    torch float64, WITHOUT the L2 term (that one is args.reg * sum of squares of the registered
    parameters, model.py:245). Returns (preLoss, sslloss, final_user, final_item).

    P: uEmbed [T,U,d], iEmbed [T,I,d], posEmbed [L,d], fusion dicts "fuse_u"/"fuse_i" (lstm_W,
       lstm_b shared), head "ln" list of (gamma, beta) and "att" list of dicts, meta2_W/b, meta3_W/b.
    batch: uids, iids, uLocs_seq, sequence [B,L], mask [B,L], suids[k], siids[k] (lists), drop_u /
       drop_i (output-dropout scales [N,T,d] or None).
    Quirks kept: the halves of every score vector are taken by POSITION (model.py:192-195, 200-201,
    242-243) although sampleSslBatch interleaves positives and negatives (model.py:331-335)."""
    import torch
    T, L, leaky, heads = cfg["T"], cfg["L"], cfg["leaky"], cfg["heads"]

    def lk(x):
        a = leaky * x
        return torch.where(a >= x, a, x)

    uv, iv = [], []
    for k in range(T):
        u, i = torch_gnn_interval(P["uEmbed"][k], P["iEmbed"][k], adj_list[k], tp_list[k], L, leaky)
        uv.append(u)
        iv.append(i)

    def fuse(x, p, drop):
        h = torch_basic_lstm(x, p["lstm_W"], p["lstm_b"])
        if drop is not None:
            h = h * drop
        return torch_mhsa_mean(torch_layer_norm_td(h, p["ln_gamma"], p["ln_beta"]), p["Wq"], p["bq"], p["Wk"],
                               p["bk"], p["Wv"], p["bv"], heads)

    fu = fuse(torch.stack(uv, 1), P["fuse_u"], batch.get("drop_u"))
    fi = fuse(torch.stack(iv, 1), P["fuse_i"], batch.get("drop_i"))
    # ---- head (model.py:156-173)
    seq = torch.as_tensor(batch["sequence"], dtype=torch.long)
    m = torch.as_tensor(batch["mask"], dtype=fu.dtype)[:, None, :]
    sb = torch_layer_norm_td(m @ fi[seq], *P["ln"][0]) + torch_layer_norm_td(
        m @ P["posEmbed"][None].expand(seq.shape[0], -1, -1), *P["ln"][1])
    att = sb
    for i, p in enumerate(P["att"]):
        a1 = torch_mhsa_mean(torch_layer_norm_td(att, *P["ln"][2 + i]), p["Wq"], p["bq"], p["Wk"], p["bk"],
                             p["Wv"], p["bv"], heads)[:, None, :]
        att = lk(a1) + att
    att_user = att.sum(1)
    uids = torch.as_tensor(batch["uids"], dtype=torch.long)
    iids = torch.as_tensor(batch["iids"], dtype=torch.long)
    ulocs = torch.as_tensor(batch["uLocs_seq"], dtype=torch.long)
    preds = (fu[uids] * fi[iids]).sum(-1) + (lk(att_user[ulocs]) * fi[iids]).sum(-1)
    n = preds.shape[0] // 2
    pre_loss = torch.clamp(1.0 - (preds[:n] - preds[n:]), min=0).mean()
    # ---- SSL (model.py:174-205)
    ssl = 0.0
    for k in range(T):
        su = torch.as_tensor(batch["suids"][k], dtype=torch.long)
        si = torch.as_tensor(batch["siids"][k], dtype=torch.long)
        meta1 = torch.cat([fu * uv[k], fu, uv[k]], dim=-1)
        meta2 = lk(meta1 @ P["meta2_W"] + P["meta2_b"])
        w = torch.sigmoid(meta2 @ P["meta3_W"] + P["meta3_b"]).squeeze(-1)[su]
        ns = su.shape[0] // 2
        s_final = lk(fu[su] * fi[si]).sum(-1).detach()
        S = w[:ns] * s_final[:ns] - w[ns:] * s_final[ns:]
        p1 = lk(uv[k][su] * iv[k][si]).sum(-1)
        ssl = ssl + torch.clamp(1.0 - S * (p1[:ns] - p1[ns:]), min=0).sum()
    return pre_loss, ssl, fu, fi
