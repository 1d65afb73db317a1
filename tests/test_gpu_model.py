"""GPU parity: the Recommender entry points (prepareModel / ours / messagePropagate) end to end
against the oracle on a synthetic dataset in the reference's on-disk structure."""
import numpy as np
import pytest
import torch

from oracle import selfgnn_oracle as O



@pytest.mark.gpu
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
    assert float(NNs.params["uEmbed"].detach().abs().max()) <= lim
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


@pytest.mark.gpu
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


@pytest.mark.gpu
def test_prediction_head_and_metrics_vs_oracle(dev):
    """Prediction head (model.py:156-173) and HR/NDCG (model.py:484-510) against the oracle,
    on a synthetic dataset with tst_int / test_dict in the reference's format."""
    import numpy as np
    from sa_gnn_amd import synthetic
    from sa_gnn_amd.DataHandler import DataHandler
    from sa_gnn_amd.Params import args
    from sa_gnn_amd.Utils import NNLayers as NNs
    from sa_gnn_amd.model import Recommender
    rng = np.random.default_rng(12)
    args.graphNum, args.gnn_layer, args.latdim, args.leaky = 3, 2, 64, 0.5
    args.att_layer, args.batch, args.pos_length, args.testSize, args.test = 2, 64, 20, 50, True
    U, I = 150, 120
    tmt = synthetic.make_trn_mat_time(U, I, [1500, 1400, 1300])
    seq = synthetic.make_sequence(tmt)
    tst_int = [int(rng.integers(0, I)) if u % 3 else None for u in range(U)]
    test_dict = {u + 1: list(rng.integers(1, I + 1, size=60)) for u in range(U)}      # 1-indexed
    handler = DataHandler.from_memory(tmt, seq, tst_int, test_dict)
    rec = Recommender(dev, handler)
    rec.prepareModel()
    g = torch.Generator(device="cpu").manual_seed(5)
    with torch.no_grad():
        for name in list(NNs.params):
            if name.endswith("bias") or name.endswith("beta"):
                NNs.params[name].copy_(0.1 * torch.randn(NNs.params[name].shape, generator=g))
        for k in ("uEmbed", "iEmbed", "posEmbed"):
            NNs.params[k].mul_(30)
    res = rec.testEpoch()
    # oracle: same batches through the numpy restatement
    cpu = lambda t: t.detach().cpu().numpy()
    fu, fi = cpu(rec.final_user_vector), cpu(rec.final_item_vector)
    ln_params = [(cpu(gm), cpu(bt)) for gm, bt in rec.head_ln]
    att_params = [{k: cpu(v) for k, v in mh.weights().items()} for mh in rec.multihead_self_attention_sequence]
    ids = handler.tstUsrs
    tot = np.zeros(6)
    for st in range(0, len(ids), args.batch):
        bat = ids[st:st + args.batch]
        uL, iL, temTst, tstLocs, sequence, mask, uLs, _ = rec.sampleTestBatch(bat)
        want = O.prediction_head(fu, fi, cpu(NNs.params["posEmbed"]), ln_params, att_params, np.array(uL), np.array(iL),
                                 sequence, mask, np.array(uLs), 16, 0.5)
        got = rec.predict(uL, iL, sequence, mask, uLs).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
        o = O.calc_res(want.reshape(len(bat), -1), temTst, tstLocs, shoot=10)
        mine = rec.calcRes(want.reshape(len(bat), -1), temTst, tstLocs, shoot=10)
        np.testing.assert_allclose(mine, o, rtol=1e-12)
        tot += np.array(o)
    assert abs(res["HR"] - tot[0] / len(ids)) <= 2.0 / len(ids) and 0 < res["HR"] <= 1
    assert abs(res["NDCG"] - tot[1] / len(ids)) <= 2.0 / len(ids)


def test_calc_res_tie_order():
    """Ties: the positive is the last candidate and a stable descending sort keeps it last."""
    from sa_gnn_amd.model import Recommender
    preds = np.array([[1.0, 1.0, 1.0, 0.5], [0.1, 0.9, 0.9, 0.9]])
    tst_locs = [np.array([7, 8, 9, 3]), np.array([4, 5, 6, 2])]
    got = Recommender.calcRes(preds, [9, 2], tst_locs, shoot=2)
    want = O.calc_res(preds, [9, 2], tst_locs, shoot=2)
    assert got == want and got[0] == 0.0          # neither positive makes the top-2
