"""GPU parity of the full training objective: loss values and the gradient of
every trainable tensor of the model against float64 torch.autograd over the oracle's restatement
of model.py:104-205, 241-246, on a batch drawn by the package's own samplers."""
import numpy as np
import pytest
import torch

from oracle import selfgnn_oracle as O

pytestmark = pytest.mark.gpu


def _setup(dev, d, ssldim, att_layer):
    from sa_gnn_amd import synthetic
    from sa_gnn_amd.DataHandler import DataHandler
    from sa_gnn_amd.Params import args
    from sa_gnn_amd.Utils import NNLayers as NNs
    from sa_gnn_amd.model import Recommender
    rng = np.random.default_rng(31)
    args.graphNum, args.gnn_layer, args.latdim, args.leaky, args.ssldim = 2, 2, d, 0.5, ssldim
    args.att_layer, args.batch, args.pos_length, args.testSize, args.test = att_layer, 16, 12, 20, True
    args.sslNum, args.pred_num, args.keepRate, args.ssl_reg, args.reg = 3, 2, 1.0, 0.5, 1e-2
    U, I = 70, 60
    tmt = synthetic.make_trn_mat_time(U, I, [700, 650])
    seq = synthetic.make_sequence(tmt)
    tst_int = [int(rng.integers(0, I)) if u % 2 else None for u in range(U)]
    handler = DataHandler.from_memory(tmt, seq, tst_int, {u + 1: list(rng.integers(1, I + 1, size=30)) for u in range(U)})
    rec = Recommender(dev, handler)
    rec.prepareModel()
    g = torch.Generator(device="cpu").manual_seed(9)
    with torch.no_grad():
        for name in list(NNs.params):
            if name.endswith("bias") or name.endswith("beta") or name.endswith("Bias"):
                NNs.params[name].copy_(0.1 * torch.randn(NNs.params[name].shape, generator=g))
        for k in ("uEmbed", "iEmbed", "posEmbed"):
            NNs.params[k].mul_(20)
    return rec, handler, NNs, args


def _oracle_params(rec, NNs):
    t64 = lambda v: v.detach().cpu().double().requires_grad_(True)
    leaves = {}

    def leaf(name):
        if name not in leaves:
            leaves[name] = t64(NNs.params[name])
        return leaves[name]

    def mh(att):
        inv = {id(v): k for k, v in NNs.params.items()}
        return {k: leaf(inv[id(v)]) for k, v in att.weights().items()}

    inv = {id(v): k for k, v in NNs.params.items()}
    P = {"uEmbed": leaf("uEmbed"), "iEmbed": leaf("iEmbed"), "posEmbed": leaf("posEmbed"),
         "meta2_W": leaf("meta2"), "meta2_b": leaf("meta2Bias"), "meta3_W": leaf("meta3"), "meta3_b": leaf("meta3Bias")}
    for key, (gm, bt), att in (("fuse_u", rec.ln[0], rec.multihead_self_attention0),
                                ("fuse_i", rec.ln[1], rec.multihead_self_attention1)):
        p = {"lstm_W": leaf("rnn_lstm_kernel"), "lstm_b": leaf("rnn_lstm_bias"), "ln_gamma": leaf(inv[id(gm)]),
             "ln_beta": leaf(inv[id(bt)])}
        p.update(mh(att))
        P[key] = p
    P["ln"] = [(leaf(inv[id(gm)]), leaf(inv[id(bt)])) for gm, bt in rec.head_ln]
    P["att"] = [mh(a) for a in rec.multihead_self_attention_sequence]
    return P, leaves


@pytest.mark.parametrize("d,ssldim,att_layer,dropout", [(64, 48, 2, False), (32, 32, 1, True)])
def test_training_objective_gradients(dev, d, ssldim, att_layer, dropout):
    rec, handler, NNs, args = _setup(dev, d, ssldim, att_layer)
    np.random.seed(3)
    import random
    random.seed(3)
    batIds = np.random.permutation(args.user)[:args.batch]
    uL, iL, sequence, mask, uLs = rec.sampleTrainBatch(batIds, handler.trnMat, handler.timeMat, 5)
    su, si, _ = rec.sampleSslBatch(batIds, handler.subMat, False)
    assert len(uL) == len(iL) == len(uLs) and len(uL) % 2 == 0 and all(len(a) % 2 == 0 for a in su)
    batch = {"uids": uL, "iids": iL, "uLocs_seq": uLs, "sequence": sequence, "mask": mask, "suids": su, "siids": si}
    obatch = dict(batch)
    if dropout:
        g = torch.Generator(device="cpu").manual_seed(4)
        for key, n in (("drop_u", args.user), ("drop_i", args.item)):
            m = (torch.rand((n, args.graphNum, d), generator=g) < 0.5).float() * 2.0
            batch[key] = m.to(dev)
            obatch[key] = m.double()
    # HIP path
    for p in NNs.params.values():
        p.grad = None
    pre, ssl = rec.train_loss(batch, keep_rate=1.0)
    (pre + args.ssl_reg * ssl).backward()
    # oracle
    P, leaves = _oracle_params(rec, NNs)
    adj = [O.trans_to_lsts(m)[0] for m in handler.subMat]
    tp = [O.trans_to_lsts(O.transpose(m))[0] for m in handler.subMat]
    opre, ossl, _, _ = O.torch_train_loss(P, adj, tp, obatch, {"T": 2, "L": 2, "leaky": 0.5, "heads": 16})
    (opre + args.ssl_reg * ossl).backward()
    assert abs(float(pre.detach()) - float(opre.detach())) <= 1e-4 * max(abs(float(opre)), 1.0)
    assert abs(float(ssl.detach()) - float(ossl.detach())) <= 1e-4 * max(abs(float(ossl)), 1.0)
    checked = 0
    for name, leaf in leaves.items():
        got = NNs.params[name].grad
        want = leaf.grad
        if want is None:
            assert got is None or float(got.abs().max()) == 0.0, name
            continue
        assert got is not None, f"no gradient reached {name}"
        a, b = got.cpu().double().numpy(), want.numpy()
        floor = max(5e-5 * np.abs(b).max(), 2e-5)
        if name.endswith("k_bias"):
            # analytically ~0 (a key bias shifts every score of a row alike); what is left is the
            # cancellation noise of terms as large as those of the key kernel's gradient
            floor = max(floor, 1e-3 * float(leaves[name.replace("k_bias", "k_kernel")].grad.abs().max()))
        tol = 2e-4 * np.abs(b) + floor
        bad = np.abs(a - b) > tol
        assert not bad.any(), f"{name}: {bad.sum()}/{bad.size} off, worst {np.abs(a - b)[bad].max():.3e} (scale {np.abs(b).max():.3e})"
        checked += 1
    assert checked >= 20
    # the dead [d, d] weights of messagePropagate and timeEmbed get no gradient (L2 only)
    assert NNs.params["timeEmbed"].grad is None


def test_train_epoch_runs_and_improves_loss(dev):
    """Two epochs of the reference's loop (samplers -> loss -> backward -> Adam) on a toy dataset:
    finite losses, parameters move, preLoss goes down; then the evaluator runs on the result."""
    rec, handler, NNs, args = _setup(dev, 64, 32, 1)
    args.trnNum, args.lr, args.keepRate, args.ssl_reg, args.reg = 64, 5e-3, 0.5, 1e-3, 1e-4
    args.decay_step = args.trnNum // args.batch
    np.random.seed(0)
    torch.manual_seed(0)                                   # dropout masks: the run is reproducible
    before = NNs.params["uEmbed"].detach().clone()
    losses = [rec.trainEpoch()["preLoss"] for _ in range(8)]
    # one 64-user step per epoch with keepRate 0.5: single steps are noisy, the trend is not
    assert all(np.isfinite(losses)) and min(losses[-3:]) < losses[0] and np.mean(losses[-3:]) < np.mean(losses[:3])
    assert float((NNs.params["uEmbed"].detach() - before).abs().max()) > 0
    res = rec.testEpoch()
    assert 0.0 <= res["HR"] <= 1.0 and 0.0 <= res["NDCG"] <= 1.0


def test_checkpoint_round_trip_resumes_identically(dev, tmp_path):
    """reference model.py:44-46, 512-526: saveHistory -> a fresh Recommender in the same process
    (a second prepareModel: variable names must come out the same) -> loadModel -> identical
    testEpoch() metrics, the resumed start epoch of run(), and an identical NEXT training step
    (Adam moments, bias correction and the lr staircase continue: tf.train.Saver stores the slots
    and globalStep with the variables)."""
    rec, handler, NNs, args = _setup(dev, 64, 32, 1)
    args.keepRate, args.trnNum, args.epoch, args.tstEpoch, args.lr, args.decay = 1.0, 32, 9, 3, 1e-2, 0.8
    args.decay_step = args.trnNum // args.batch
    args.save_path, args.load_model = "ckpt_test", None
    np.random.seed(5)
    for ep in range(4):                                   # 4 epochs x 2 steps: the staircase has moved
        rec.makePrint("Train", ep, rec.trainEpoch(), ep % args.tstEpoch == 0)
        if ep % args.tstEpoch == 0:
            rec.makePrint("Test", ep, rec.testEpoch(), True)
    assert rec.optimizer.global_step == 8 and rec.optimizer.learning_rate() < args.lr
    want = rec.testEpoch()
    rec.saveHistory(str(tmp_path))
    saved_step = rec.optimizer.global_step
    names = sorted(NNs.params)
    # the step the ORIGINAL model takes next, on a fixed batch
    state = np.random.get_state()
    before = {k: v.detach().clone() for k, v in NNs.params.items()}
    loss_a = rec.trainEpoch()
    after_a = {k: v.detach().clone() for k, v in NNs.params.items()}
    assert max(float((after_a[k] - before[k]).abs().max()) for k in before) > 1e-3      # the step is visible

    from sa_gnn_amd.model import Recommender
    rec2 = Recommender(dev, handler)
    rec2.prepareModel()                                   # fresh registry, fresh random init
    assert sorted(NNs.params) == names                    # attention variables keep their names
    args.load_model = "ckpt_test"
    rec2.loadModel(str(tmp_path))
    assert rec2.optimizer.global_step == saved_step
    assert len(rec2.metrics["TrainLoss"]) * args.tstEpoch - (args.tstEpoch - 1) == 4     # stloc of run()
    got = rec2.testEpoch()
    assert got == want
    np.random.set_state(state)
    loss_b = rec2.trainEpoch()
    # weight-gradient sums use float atomics (last-bit run-to-run differences), hence not torch.equal;
    # a step from reset moments or a restarted staircase would differ by ~lr = 4e-3
    assert loss_b == pytest.approx(loss_a, rel=1e-5)
    for k, v in NNs.params.items():
        head_qk = k.startswith("mhsa") and int(k[4:k.index("_")]) >= 3 and k.split("_")[1] in ("q", "k")
        if k.endswith("k_bias") or head_qk:
            # gradients that are analytically ~0: a key bias shifts every score of a row alike, and the
            # head's attention runs on a length-1 sequence (attn = e / (e + 1e-8) ~ 1 whatever Q and K are).
            # What is left is last-bit summation noise, which Adam's m / sqrt(v) normalises to +-lr:
            # not reproducible between two runs of the same step
            continue
        torch.testing.assert_close(v.detach(), after_a[k], rtol=1e-4, atol=2e-5, msg=k)
    # a checkpoint that does not fit the model is refused
    args.latdim = 32
    rec3 = Recommender(dev, handler)
    rec3.prepareModel()
    with pytest.raises(ValueError, match="shape"):
        rec3.loadModel(str(tmp_path))
    args.load_model, args.latdim = None, 64


def test_adam_decays_registered_tensors_without_gradient(dev):
    """timeEmbed and the dead [d, d] weights (model.py:81, :117) get no gradient from the forward
    ops but are in regParams: minimize(loss + reg*Regularize()) moves them by the L2 term alone.
    One multi-tensor launch; un-regularised tensors without a gradient stay put."""
    from sa_gnn_amd import ops
    rng = np.random.default_rng(4)
    w0, t0, b0 = (rng.standard_normal(s).astype(np.float32) for s in ((37, 5), (2, 64), (1003,)))
    params = {"w": torch.from_numpy(w0.copy()).to(dev), "timeEmbed": torch.from_numpy(t0.copy()).to(dev),
              "bias": torch.from_numpy(b0.copy()).to(dev)}
    opt = ops.Adam(params, lr=1e-2, decay=0.9, decay_step=2, reg=1e-2, reg_names={"w", "timeEmbed"})
    ref = {k: [v.astype(np.float64), np.zeros(v.shape), np.zeros(v.shape)] for k, v in (("w", w0), ("timeEmbed", t0))}
    for step in range(1, 5):
        g = rng.standard_normal(w0.shape).astype(np.float32)
        opt.step({"w": torch.from_numpy(g).to(dev), "timeEmbed": None, "bias": None})
        lr = 1e-2 * 0.9 ** ((step - 1) // 2)
        for k, grad in (("w", g.astype(np.float64)), ("timeEmbed", 0.0)):
            p, m, v = ref[k]
            gg = grad + 2 * 1e-2 * p
            m[:] = 0.9 * m + 0.1 * gg
            v[:] = 0.999 * v + 0.001 * gg * gg
            p -= lr * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step) * m / (np.sqrt(v) + 1e-8)
    np.testing.assert_allclose(params["w"].cpu().numpy(), ref["w"][0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(params["timeEmbed"].cpu().numpy(), ref["timeEmbed"][0], rtol=1e-4, atol=1e-6)
    assert not np.array_equal(params["timeEmbed"].cpu().numpy(), t0)
    np.testing.assert_array_equal(params["bias"].cpu().numpy(), b0)
