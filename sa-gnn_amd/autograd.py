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
