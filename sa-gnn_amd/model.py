"""Recommender with the reference's entry points (reference model.py:18-250) over libsagnn.so.

What is kept: `Recommender(sess, handler)`, `.prepareModel()`, `.ours()`,
`.messagePropagate(srclats, mat, type)`, `.edgeDropout(mat)`, the parameter names/shapes and the
L2 registry. `sess` is the device context (a torch.device or its string); the TF graph/session
split disappears, so `ours()` runs the hot path eagerly on the current HIP stream.

Scope: the per-interval propagation stack and the interval fusion — everything
that produces `final_user_vector` / `final_item_vector` (reference model.py:104-155). The
prediction head, SSL loss, samplers and optimiser around it are §8(f) "next" rows.
"""
from __future__ import annotations

import torch

from . import ops
from .Params import args
from .Utils import NNLayers as NNs
from .Utils.attention import MultiHeadSelfAttention
from .graph import interval_pair


def random_fusion_params(d: int, device, seed: int = 0) -> dict:
    """Random-init fusion parameters with the shapes TF creates (BasicLSTMCell kernel [2d, 4d] and
    bias [4d]; layer_norm gamma/beta [d]; three dense kernels [d, d] with bias [d]). Kernels are
    xavier-uniform; biases/beta get small random values so benchmarks and tests exercise them."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def xavier(r, c):
        lim = (6.0 / (r + c)) ** 0.5
        return (torch.rand((r, c), generator=g) * 2 * lim - lim).to(device)

    def small(n, mean=0.0):
        return (mean + 0.1 * torch.randn(n, generator=g)).to(device)

    return {"lstm_W": xavier(2 * d, 4 * d), "lstm_b": small(4 * d), "ln_gamma": small(d, 1.0),
            "ln_beta": small(d), "Wq": xavier(d, d), "bq": small(d), "Wk": xavier(d, d), "bk": small(d),
            "Wv": xavier(d, d), "bv": small(d)}


class Recommender:
    def __init__(self, sess, handler):
        self.sess = sess
        self.device = torch.device(sess if sess is not None else "cuda:0")
        self.handler = handler
        print("USER", args.user, "ITEM", args.item)
        self.metrics = dict()
        for met in ["Loss", "preLoss", "HR", "NDCG"]:
            self.metrics["Train" + met] = list()
            self.metrics["Test" + met] = list()

    # ------------------------------------------------------------------ hot-path pieces
    def messagePropagate(self, srclats, mat, type="user"):
        """reference model.py:80-92. `mat` is an IntervalAdj whose rows are the target nodes;
        `type` selected the row count in the reference (self.users / self.items) and is implied
        by mat.dense_shape[0] here. Registers the same dead, L2-regularised [d, d] weight the
        reference creates per call (FC(self.timeEmbed, latdim, reg=True), model.py:81)."""
        NNs.defineRandomNameParam([args.latdim, args.latdim], reg=True)
        expect = args.user if type == "user" else args.item
        if mat.dense_shape[0] != expect:
            raise ValueError(f"type={type!r} expects {expect} target rows, adjacency has {mat.dense_shape[0]}")
        return ops.spmm(mat.plan, srclats.detach(), NNs.leaky)

    def edgeDropout(self, mat):
        """reference model.py:93-102 rewrites edge VALUES only; messagePropagate never reads them
        (model.py:84-86), so the forward result is independent of keepRate and TF prunes the op.
        Identity here."""
        return mat

    def _define_fusion_params(self):
        d = args.latdim
        # tf.contrib.rnn.BasicLSTMCell(d) shared by users and items (model.py:135-144):
        # kernel [2d, 4d] glorot-uniform (TF's default initializer), bias zeros.
        self.lstm_kernel = NNs.defineParam("rnn_lstm_kernel", [2 * d, 4 * d])
        self.lstm_bias = NNs.defineParam("rnn_lstm_bias", [4 * d], initializer="zeros")
        # two layer_norm calls -> separate gamma/beta (model.py:152-153)
        self.ln = []
        for tag in ("LayerNorm", "LayerNorm_1"):
            self.ln.append((NNs.defineParam(tag + "_gamma", [d], initializer="ones"),
                            NNs.defineParam(tag + "_beta", [d], initializer="zeros")))
        self.multihead_self_attention0 = MultiHeadSelfAttention(d, args.num_attention_heads)
        self.multihead_self_attention1 = MultiHeadSelfAttention(d, args.num_attention_heads)

    def propagate_intervals(self, intervals=None):
        """reference model.py:118-134: for every interval k the L-layer stack with residuals and
        add_n, written straight into [N, T, d] slabs (no stack/transpose pass). `intervals`
        restricts the loop to a rank's shard (parallel.py); other columns are left untouched."""
        T, d, L = args.graphNum, args.latdim, args.gnn_layer
        if self.user_vector_tensor is None:
            self.user_vector_tensor = torch.empty((args.user, T, d), dtype=torch.float32, device=self.device)
            self.item_vector_tensor = torch.empty((args.item, T, d), dtype=torch.float32, device=self.device)
        if L > 1 and self._scratch_u is None:
            self._scratch_u = torch.empty((2, args.user, d), dtype=torch.float32, device=self.device)
            self._scratch_i = torch.empty((2, args.item, d), dtype=torch.float32, device=self.device)
        for k in (range(T) if intervals is None else intervals):
            if L == 0:
                self.user_vector_tensor[:, k, :].copy_(self.uEmbed[k].detach())
                self.item_vector_tensor[:, k, :].copy_(self.iEmbed[k].detach())
                continue
            ops.gnn_interval(self.subAdj[k].plan, self.subTpAdj[k].plan, self.uEmbed[k].detach(),
                             self.iEmbed[k].detach(), L, NNs.leaky,
                             self.user_vector_tensor[:, k, :], self.item_vector_tensor[:, k, :],
                             self._scratch_u, self._scratch_i)
        return self.user_vector_tensor, self.item_vector_tensor

    def fuse_intervals(self, user_vector_tensor, item_vector_tensor):
        """reference model.py:135-155: shared LSTM, per-type layer_norm + MHSA, mean over T."""
        heads = args.num_attention_heads
        outs = []
        for x, (gamma, beta), att in ((user_vector_tensor, self.ln[0], self.multihead_self_attention0),
                                      (item_vector_tensor, self.ln[1], self.multihead_self_attention1)):
            p = {"lstm_W": self.lstm_kernel.detach(), "lstm_b": self.lstm_bias.detach(),
                 "ln_gamma": gamma.detach(), "ln_beta": beta.detach()}
            p.update({k: v.detach() for k, v in att.weights().items()})
            outs.append(ops.interval_fusion(x, p, heads))
        return outs[0], outs[1]

    def ours(self):
        """The hot path of reference model.py:104-155. Returns (final_user_vector [U, d],
        final_item_vector [I, d]); the reference's (preds, sslloss) are built on top of these by
        the head / SSL branch (model.py:156-205), outside this build's scope."""
        T, d = args.graphNum, args.latdim
        self.uEmbed = NNs.defineParam("uEmbed", [T, args.user, d], reg=True)
        self.iEmbed = NNs.defineParam("iEmbed", [T, args.item, d], reg=True)
        self.posEmbed = NNs.defineParam("posEmbed", [args.pos_length, d], reg=True)
        self.timeEmbed = NNs.defineParam("timeEmbed", [self.maxTime + 1, d], reg=True)
        # one dead [d, d] weight per messagePropagate call: 2*T*L of them (model.py:81, :122-123)
        for _ in range(2 * T * args.gnn_layer):
            NNs.defineRandomNameParam([d, d], reg=True)
        self._define_fusion_params()
        return self.forward()

    def forward(self):
        """Re-runs the hot path with the current parameters (what every sess.run recomputes)."""
        uvt, ivt = self.propagate_intervals()
        self.final_user_vector, self.final_item_vector = self.fuse_intervals(uvt, ivt)
        return self.final_user_vector, self.final_item_vector

    def capture_forward(self, warmup: int = 2):
        """Captures forward() into a hipGraph (torch.cuda.CUDAGraph) and returns a replay callable.
        Real datasets are launch-bound on MI355X (each SpMM is tens of microseconds): one graph
        launch replaces 2*T*L + 4 kernel launches. Outputs land in the same tensors every replay
        (self.final_user_vector / self.final_item_vector); parameters are read in place."""
        for _ in range(max(warmup, 1)):      # first call configures kernels / allocates workspaces
            self.forward()
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.forward()

        def replay():
            graph.replay()
            return self.final_user_vector, self.final_item_vector

        self._graph = graph
        return replay

    # ------------------------------------------------------------------ model construction
    def prepareModel(self):
        """reference model.py:207-240 up to the call of ours(): adjacency constants for every
        interval and both directions, leaky slope, then the hot path."""
        NNs.reset(self.device)
        NNs.leaky = args.leaky
        self.actFunc = "leakyRelu"
        self.subAdj, self.subTpAdj = [], []
        for i in range(args.graphNum):
            adj, tp = interval_pair(self.handler.subMat[i], self.device)
            self.subAdj.append(adj)
            self.subTpAdj.append(tp)
        self.maxTime = self.handler.maxTime
        self.user_vector_tensor = self.item_vector_tensor = None
        self._scratch_u = self._scratch_i = None
        self.final_user_vector, self.final_item_vector = self.ours()

    def run(self):
        self.prepareModel()
        return self.final_user_vector, self.final_item_vector
