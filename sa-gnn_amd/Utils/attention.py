"""MultiHeadSelfAttention with the reference's constructor and parameters
(reference Utils/attention.py:47-78): three dense layers WITH bias, xavier-uniform kernels, no
output projection, scores = exp(QK^T/sqrt(d_k)) normalised by (sum + 1e-8).
The arithmetic is sagnn_mhsa_mean_f32 (attention + the caller's reduce_mean, model.py:154-155)."""
from __future__ import annotations

from .. import ops
from . import NNLayers as NNs


class MultiHeadSelfAttention:

    def __init__(self, d_model, num_attention_heads):
        assert d_model % num_attention_heads == 0                     # reference attention.py:51
        self.d_model = d_model
        self.num_attention_heads = num_attention_heads
        self.d_k = self.d_v = d_model // num_attention_heads
        tag = "mhsa%d_" % NNs.getMhsaId()
        # tf.layers.dense variables: kernel [d, d] xavier, bias [d] zeros; not L2-regularised
        self.Wq = NNs.defineParam(tag + "q_kernel", [d_model, d_model])
        self.bq = NNs.defineParam(tag + "q_bias", [d_model], initializer="zeros")
        self.Wk = NNs.defineParam(tag + "k_kernel", [d_model, d_model])
        self.bk = NNs.defineParam(tag + "k_bias", [d_model], initializer="zeros")
        self.Wv = NNs.defineParam(tag + "v_kernel", [d_model, d_model])
        self.bv = NNs.defineParam(tag + "v_bias", [d_model], initializer="zeros")

    def weights(self):
        return {"Wq": self.Wq, "bq": self.bq, "Wk": self.Wk, "bk": self.bk, "Wv": self.Wv, "bv": self.bv}

    def attention_mean(self, Q, out=None):
        """reduce_mean(attention(Q), axis=1): [N, T, d] -> [N, d]."""
        return ops.mhsa_mean(Q.detach(), self.Wq.detach(), self.bq.detach(), self.Wk.detach(),
                             self.bk.detach(), self.Wv.detach(), self.bv.detach(),
                             self.num_attention_heads, out=out)
