"""GPU parity: the Recommender entry points (prepareModel / ours / messagePropagate) end to end
against the oracle on a synthetic dataset in the reference's on-disk structure."""
import numpy as np
import pytest
import torch

from oracle import selfgnn_oracle as O

pytestmark = pytest.mark.gpu


def test_recommender_hot_path_vs_oracle(dev):
    from sa_gnn_amd import synthetic
    from sa_gnn_amd.DataHandler import DataHandler
    from sa_gnn_amd.Params import args
    from sa_gnn_amd.Utils import NNLayers as NNs
    from sa_gnn_amd.model import Recommender
    args.graphNum, args.gnn_layer, args.latdim, args.leaky = 4, 2, 64, 0.5
    U, I = 500, 330
    tmt = synthetic.make_trn_mat_time(U, I, [4000, 3500, 0, 2500])     # interval 2 is empty
    handler = DataHandler.from_memory(tmt, synthetic.make_sequence(tmt))
    rec = Recommender(dev, handler)
    rec.prepareModel()
    assert rec.final_user_vector.shape == (U, 64) and rec.final_item_vector.shape == (I, 64)
    # the registry matches the reference: 2*T*L dead [d,d] weights + uEmbed/iEmbed/posEmbed/timeEmbed
    assert sum(1 for k in NNs.regParams if k.startswith("defaultParamName")) == 2 * 4 * 2
    assert NNs.params["uEmbed"].shape == (4, U, 64) and NNs.params["timeEmbed"].shape == (2, 64)
    lim = np.sqrt(6.0 / (4 * U + 4 * 64))
    assert float(NNs.params["uEmbed"].abs().max()) <= lim
    # give the zero-initialised biases / beta non-trivial values, then recompute
    g = torch.Generator(device="cpu").manual_seed(1)
    with torch.no_grad():
        for name in list(NNs.params):
            if name.endswith("bias") or name.endswith("beta"):
                NNs.params[name].copy_(0.1 * torch.randn(NNs.params[name].shape, generator=g))
        NNs.params["uEmbed"].mul_(30)
        NNs.params["iEmbed"].mul_(30)
    fu, fi = rec.forward()
    adjs = [O.trans_to_lsts(m)[0] for m in handler.subMat]
    tps = [O.trans_to_lsts(O.transpose(m))[0] for m in handler.subMat]
    ue, ie = NNs.params["uEmbed"].detach().cpu().numpy(), NNs.params["iEmbed"].detach().cpu().numpy()
    uv, iv = O.gnn_stack(ue, ie, adjs, tps, 2, 0.5)
    np.testing.assert_allclose(rec.user_vector_tensor.cpu().numpy(), uv, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rec.item_vector_tensor.cpu().numpy(), iv, rtol=1e-4, atol=1e-5)
    cpu = lambda t: t.detach().cpu().numpy()
    for x, (gamma, beta), att, got in ((uv, rec.ln[0], rec.multihead_self_attention0, fu),
                                        (iv, rec.ln[1], rec.multihead_self_attention1, fi)):
        p = {"lstm_W": cpu(rec.lstm_kernel), "lstm_b": cpu(rec.lstm_bias), "ln_gamma": cpu(gamma), "ln_beta": cpu(beta)}
        p.update({k: cpu(v) for k, v in att.weights().items()})
        np.testing.assert_allclose(got.cpu().numpy(), O.interval_fusion(x, p, 16), rtol=1e-4, atol=2e-5)
    # messagePropagate keeps the reference's signature and type check
    y = rec.messagePropagate(NNs.params["iEmbed"][0], rec.edgeDropout(rec.subAdj[0]), "user")
    want = O.message_propagate_zero_fill(ie[0], adjs[0], U, 0.5)
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-4, atol=1e-5)
    with pytest.raises(ValueError):
        rec.messagePropagate(NNs.params["iEmbed"][0], rec.subAdj[0], "item")


def test_graph_replay_matches_eager(dev):
    """hipGraph capture of the whole hot path: replays track parameter updates and match eager."""
    from sa_gnn_amd import synthetic
    from sa_gnn_amd.DataHandler import DataHandler
    from sa_gnn_amd.Params import args
    from sa_gnn_amd.Utils import NNLayers as NNs
    from sa_gnn_amd.model import Recommender
    args.graphNum, args.gnn_layer, args.latdim, args.leaky = 3, 2, 64, 0.5
    tmt = synthetic.make_trn_mat_time(400, 300, [3000, 2500, 2800])
    rec = Recommender(dev, DataHandler.from_memory(tmt, synthetic.make_sequence(tmt)))
    rec.prepareModel()
    replay = rec.capture_forward()
    with torch.no_grad():
        NNs.params["uEmbed"].mul_(25)
        NNs.params["iEmbed"].mul_(25)
    fu_g, fi_g = (t.clone() for t in replay())
    fu_e, fi_e = rec.forward()
    torch.testing.assert_close(fu_g, fu_e, rtol=0, atol=0)
    torch.testing.assert_close(fi_g, fi_e, rtol=0, atol=0)
