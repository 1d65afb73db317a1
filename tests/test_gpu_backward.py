"""GPU parity of the backward pass of the interval stack against
torch.autograd over the oracle's torch restatement (CPU, float64)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import selfgnn_oracle as O

pytestmark = pytest.mark.gpu


def _case(rng, U, I, dens, d):
    m = sp.csr_matrix((rng.random((U, I)) < dens).astype(np.intc))
    u0 = rng.standard_normal((U, d)).astype(np.float32)
    i0 = rng.standard_normal((I, d)).astype(np.float32)
    gu = rng.standard_normal((U, d)).astype(np.float32)
    gi = rng.standard_normal((I, d)).astype(np.float32)
    return m, u0, i0, gu, gi


@pytest.mark.parametrize("d,L,tuning", [(64, 2, None), (32, 1, (4, 8, 64)), (128, 3, (8, 16, 64)), (64, 3, (2, 4, 64))])
def test_gnn_interval_backward_vs_autograd(dev, d, L, tuning):
    from sa_gnn_amd import graph, ops
    rng = np.random.default_rng(d * 10 + L)
    U, I = 157, 211
    m, u0, i0, gu, gi = _case(rng, U, I, 0.07, d)
    m = m.tolil()
    m[5, :] = 0                      # an isolated user: s = 0 exactly, the tie case of tf.maximum
    m = sp.csr_matrix(m)
    adj_idx, tp_idx = O.trans_to_lsts(m)[0], O.trans_to_lsts(O.transpose(m))[0]
    # oracle: torch autograd in float64
    tu = torch.tensor(u0, dtype=torch.float64, requires_grad=True)
    ti = torch.tensor(i0, dtype=torch.float64, requires_grad=True)
    ou, oi = O.torch_gnn_interval(tu, ti, adj_idx, tp_idx, L, 0.5)
    (ou * torch.tensor(gu, dtype=torch.float64)).sum().add((oi * torch.tensor(gi, dtype=torch.float64)).sum()).backward()
    # HIP path
    fwd, tp = graph.interval_pair(m, dev, tuning=tuning)
    mask_u = torch.empty((L, U, d // 4), dtype=torch.uint8, device=dev)
    mask_i = torch.empty((L, I, d // 4), dtype=torch.uint8, device=dev)
    uo = torch.empty((U, d), device=dev)
    io = torch.empty((I, d), device=dev)
    ops.gnn_interval(fwd.plan, tp.plan, torch.from_numpy(u0).to(dev), torch.from_numpy(i0).to(dev), L, 0.5, uo, io,
                     mask_u=mask_u, mask_i=mask_i)
    np.testing.assert_allclose(uo.cpu().numpy(), ou.detach().numpy(), rtol=1e-4, atol=1e-4)
    du, di = ops.gnn_interval_bwd(fwd.plan, tp.plan, torch.from_numpy(gu).to(dev), torch.from_numpy(gi).to(dev),
                                  L, 0.5, mask_u, mask_i)
    scale = max(float(tu.grad.abs().max()), 1.0)
    np.testing.assert_allclose(du.cpu().numpy(), tu.grad.numpy(), rtol=1e-4, atol=2e-6 * scale * L * 50)
    np.testing.assert_allclose(di.cpu().numpy(), ti.grad.numpy(), rtol=1e-4, atol=2e-6 * scale * L * 50)


def test_autograd_function_end_to_end(dev):
    """GnnIntervalFn inside a torch graph: gradients of a scalar loss w.r.t. the embedding tables."""
    from sa_gnn_amd import autograd as ag
    from sa_gnn_amd import graph
    rng = np.random.default_rng(3)
    U, I, d, L = 90, 120, 64, 2
    m, u0, i0, _, _ = _case(rng, U, I, 0.1, d)
    fwd, tp = graph.interval_pair(m, dev)
    pu = torch.from_numpy(u0).to(dev).requires_grad_(True)
    pi = torch.from_numpy(i0).to(dev).requires_grad_(True)
    uo, io = ag.gnn_interval(pu, pi, fwd.plan, tp.plan, L, 0.5)
    loss = (uo ** 2).sum() * 0.5 + (io[:, :8] * 3.0).sum()
    loss.backward()
    tu = torch.tensor(u0, dtype=torch.float64, requires_grad=True)
    ti = torch.tensor(i0, dtype=torch.float64, requires_grad=True)
    ou, oi = O.torch_gnn_interval(tu, ti, O.trans_to_lsts(m)[0], O.trans_to_lsts(O.transpose(m))[0], L, 0.5)
    ((ou ** 2).sum() * 0.5 + (oi[:, :8] * 3.0).sum()).backward()
    s = float(tu.grad.abs().max())
    np.testing.assert_allclose(pu.grad.cpu().numpy(), tu.grad.numpy(), rtol=2e-4, atol=1e-5 * s)
    np.testing.assert_allclose(pi.grad.cpu().numpy(), ti.grad.numpy(), rtol=2e-4, atol=1e-5 * s)
