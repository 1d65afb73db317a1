"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/sagnn.h declares,
the host-side logic behind it (plan chunking, CSR validation, argument errors) and the Python
host mirror (Params, DataHandler contract, synthetic writer). No GPU compute is called."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import ROOT
from sa_gnn_amd import _lib
from sa_gnn_amd.ops import SpmmPlan


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "sagnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sagnn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    names = _header_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in sagnn.h but not exported by libsagnn.so"
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES table out of sync with sagnn.h"
    assert lib.sagnn_version() == 10302


def test_csr_check_errors():
    lib = _lib.load()
    rp = np.array([0, 2, 2, 3], np.int32)
    ci = np.array([1, 0, 2], np.int32)
    assert lib.sagnn_csr_check_host(rp.ctypes.data, ci.ctypes.data, 3, 3, 3) == 0
    bad = np.array([1, 0, 3], np.int32)
    assert lib.sagnn_csr_check_host(rp.ctypes.data, bad.ctypes.data, 3, 3, 3) == -4
    assert "colidx" in _lib.last_error()
    dec = np.array([0, 2, 1, 3], np.int32)
    assert lib.sagnn_csr_check_host(dec.ctypes.data, ci.ctypes.data, 3, 3, 3) == -4
    assert lib.sagnn_csr_check_host(rp.ctypes.data, ci.ctypes.data, 3, 3, 4) == -4
    assert lib.sagnn_csr_check_host(None, ci.ctypes.data, 3, 3, 3) == -1
    with pytest.raises(_lib.SagnnError):
        SpmmPlan(rp, bad, 3, 3)


def test_plan_chunking_host_only():
    """Long rows are cut into equal chunks (multiples of 64, <= chunk_edges), consecutive and in
    edge order; short/medium rows need no metadata."""
    rng = np.random.default_rng(0)
    deg = rng.integers(0, 40, size=500)
    deg[7] = 5000
    deg[123] = 2049
    deg[499] = 130000
    deg[300] = 2048          # == long_thresh: stays a medium row
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    colidx = np.zeros(rowptr[-1], np.int32)
    dflt = SpmmPlan(rowptr, colidx, 500, 1).info
    assert (dflt.short_thresh, dflt.long_thresh, dflt.chunk_edges) == (16, 256, 256)
    plan = SpmmPlan(rowptr, colidx, 500, 1, tuning=(16, 2048, 1024))
    info = plan.info
    assert (info.short_thresh, info.long_thresh, info.chunk_edges) == (16, 2048, 1024)
    assert info.on_device == 0 and info.max_degree == 130000
    assert info.n_long_rows == 3
    rows, e0, e1 = plan.chunks()
    assert len(rows) == info.n_chunks
    for r in (7, 123, 499):
        sel = rows == r
        b, e = e0[sel], e1[sel]
        assert b[0] == rowptr[r] and e[-1] == rowptr[r + 1]
        np.testing.assert_array_equal(b[1:], e[:-1])
        lens = e - b
        assert lens.max() <= 1024 and np.all(lens[:-1] % 64 == 0) and np.all(lens[:-1] == lens[0])
        assert len(b) == -(-deg[r] // 1024)
    assert 300 not in rows
    assert plan.workspace_bytes(64) == info.n_chunks * 64 * 4
    # custom tuning: everything above 8 edges is "long", 64-edge chunks
    plan2 = SpmmPlan(rowptr, colidx, 500, 1, tuning=(4, 8, 64))
    assert plan2.info.n_long_rows == int((deg > 8).sum())
    r2, b2, e2 = plan2.chunks()
    assert np.all(e2 - b2 <= 64)
    # spmm on a host-only plan is refused, not silently skipped
    lib = _lib.load()
    rc = lib.sagnn_spmm_f32(plan.handle, None, 0, 64, None, 0, 0.5, None, 0, None, 0, None, 0, None, 0, None)
    assert rc == -5 and "host-only" in _lib.last_error()


def test_plan_rejects_bad_rowptr():
    with pytest.raises(_lib.SagnnError):
        SpmmPlan(np.array([0, 3, 2], np.int32), np.zeros(2, np.int32), 2, 4, validate=False)
    with pytest.raises(ValueError):
        SpmmPlan(np.array([0, 1], np.int32), np.zeros(1, np.int32), 2, 4)
    with pytest.raises(TypeError):
        SpmmPlan(np.array([0, 1, 1], np.int64), np.zeros(1, np.int32), 2, 4)


def test_fusion_argument_errors_without_gpu():
    lib = _lib.load()
    assert lib.sagnn_lstm_fwd_f32(None, 0, 0, 4, 2, 64, None, None, 1.0, None, None, 0, None) == -1
    assert lib.sagnn_lstm_fwd_f32(None, 0, 0, 4, 2, 62, None, None, 1.0, None, None, 0, None) == -2
    assert lib.sagnn_mhsa_mean_f32(None, 0, 0, 4, 2, 64, 7, None, None, None, None, None, None, None, 0, None) == -2
    assert lib.sagnn_interval_fusion_workspace_bytes(10, 3, 64) == 10 * 3 * 64 * 4
    assert lib.sagnn_interval_fusion_workspace_bytes(10, 3, 128) == 10 * 3 * 128 * 4 + 10 * 3 * 3 * 128 * 4
    assert lib.sagnn_interval_fusion_workspace_bytes(10, 3, 48) == 10 * 3 * 48 * 4
    assert lib.sagnn_layernorm_td_f32(None, 0, 0, 0, 0, 64, None, None, 1e-12, None, 0, None) == -2
    # training entries: dimension and pointer checks come before any device work
    assert lib.sagnn_lstm_bwd_supported(64) == 1 and lib.sagnn_lstm_bwd_supported(128) == 0
    assert lib.sagnn_lstm_bwd_f32(None, 64, 64, None, None, None, None, 128, None, None, None, None, None, 4, 2, 128, None) == -2
    assert lib.sagnn_lstm_bwd_f32(None, 64, 64, None, None, None, None, 128, None, None, None, None, None, 4, 2, 64, None) == -1
    assert lib.sagnn_attn_bwd_front_supported(64, 2, 16) == 1
    assert lib.sagnn_attn_bwd_front_supported(64, 7, 16) == 0 and lib.sagnn_attn_bwd_front_supported(64, 2, 4) == 0
    assert lib.sagnn_attn_bwd_front_f32(None, 0, 0, 4, 7, 64, 16, None, None, 1e-12, 1, None, None, None, None, None, None,
                                        None, 64, None, None, None) == -2
    assert lib.sagnn_attn_bwd_front_f32(None, 0, 0, 4, 2, 64, 16, None, None, 1e-12, 1, None, None, None, None, None, None,
                                        None, 64, None, None, None) == -1
    assert lib.sagnn_ln_mhsa_mean_workspace_bytes(10, 3, 64, 16) == 0          # normalised in registers
    assert lib.sagnn_ln_mhsa_mean_workspace_bytes(10, 3, 128, 16) == 0         # d = 128, 16 heads: the split-bf16 fused kernel
    assert lib.sagnn_ln_mhsa_mean_workspace_bytes(10, 3, 128, 8) == 10 * 3 * 128 * 4 + 10 * 3 * 3 * 128 * 4   # wide path
    assert lib.sagnn_ln_mhsa_mean_f32(None, 0, 0, 4, 2, 64, 16, None, None, 1e-12, None, None, None, None, None, None,
                                      None, 64, None, 0, None) == -1


def test_round3_entries_reject_bad_arguments_without_gpu():
    """The entries added in round 3 validate before they touch a device: segmented weight gradient, BPTT with scratch."""
    lib = _lib.load()
    buf = (ctypes.c_float * 4096)()
    p = ctypes.addressof(buf)
    # dense_tn_seg: din / dout multiples of 32, 16-byte rows, segment strides multiples of 4 floats, dW given
    assert lib.sagnn_dense_tn_seg_f32(p, 64, 4096, p, 128, 4096, 10, 3, 48, 128, p, None, None) == -2
    assert lib.sagnn_dense_tn_seg_f32(p, 64, 4098, p, 128, 4096, 10, 3, 64, 128, p, None, None) == -3
    assert lib.sagnn_dense_tn_seg_f32(p, 62, 4096, p, 128, 4096, 10, 3, 64, 128, p, None, None) == -3
    assert lib.sagnn_dense_tn_seg_f32(p, 64, 4096, p, 128, 4096, 10, 3, 64, 128, None, None, None) == -1
    assert lib.sagnn_dense_tn_seg_f32(p, 64, 4096, p, 128, 4096, 0, 3, 64, 128, p, None, None) == 0      # nothing to add
    # lstm_bwd_ws: the scratch size is the gate gradients', a short one is an error (not a silent one-launch fallback)
    assert lib.sagnn_lstm_bwd_workspace_bytes(1000, 5, 64) == 1000 * 5 * 256 * 4
    assert lib.sagnn_lstm_bwd_workspace_bytes(0, 5, 64) == 0
    args = (p, 320, 64, p, p, p, p, 320, None, p, p, p, p, 10, 5)
    assert lib.sagnn_lstm_bwd_ws_f32(*args, 64, p, 16, None) == -6
    assert lib.sagnn_lstm_bwd_ws_f32(*args, 48, p, 1 << 20, None) == -2
    assert lib.sagnn_lstm_bwd_ws_f32(p, 320, 64, None, p, p, p, 320, None, p, p, p, p, 10, 5, 64, p, 1 << 20, None) == -1
    msg = ctypes.create_string_buffer(256)
    lib.sagnn_last_error(msg, 256)
    assert b"null" in msg.value.lower()


def test_params_match_reference_flags():
    from sa_gnn_amd import Params
    a = Params.parse_args([])
    # names/defaults of reference Params.py:5-50 used by the path
    assert (a.graphNum, a.gnn_layer, a.latdim, a.leaky, a.keepRate, a.num_attention_heads) == (8, 2, 64, 0.5, 0.5, 16)
    assert (a.batch, a.trnNum, a.decay_step, a.data, a.pos_length, a.att_layer) == (512, 10000, 19, "yelp", 200, 4)
    # the reference's shell lines parse unchanged (gowalla.sh / amazon.sh)
    g = Params.parse_args("--data gowalla --lr 2e-3 --reg 1e-2 --temp 0.1 --ssl_reg 1e-6 --save_path gowalla "
                          "--epoch 150 --batch 512 --sslNum 40 --graphNum 3 --gnn_layer 2 --att_layer 1 "
                          "--test True --testSize 1000 --ssldim 48".split())
    assert (g.graphNum, g.gnn_layer, g.data, g.test, g.ssldim) == (3, 2, "gowalla", True, 48)
    assert Params.parse_args(["--test", "False"]).test is True     # type=bool quirk (Params.py:47-49)


def test_synthetic_writer_and_datahandler_contract(tmp_path):
    from sa_gnn_amd import synthetic
    from sa_gnn_amd.DataHandler import DataHandler
    from sa_gnn_amd.Params import args
    tmt = synthetic.make_trn_mat_time(300, 200, [1500, 1200, 0], seed0=1000)
    assert len(tmt) == 3 and len(tmt[1]) == 3
    for s in tmt[1]:
        assert sp.isspmatrix_csr(s) and s.dtype == np.intc and s.shape == (300, 200)
    assert tmt[1][0].data.min() > 1_000_000_000          # values are Unix timestamps
    assert tmt[1][2].nnz == 0
    seq = synthetic.make_sequence(tmt)
    assert len(seq) == 300 and sum(len(s) for s in seq) == sum(s.nnz for s in tmt[1])
    h = DataHandler.from_memory(tmt, seq)
    assert (args.user, args.item) == (300, 200) and h.maxTime == 1
    assert h.trnMat.shape == (300, 200) and len(h.subMat) == 3
    # same objects through the reference's pickle files
    import pickle
    d = tmp_path / "synth"
    d.mkdir()
    for name, obj in (("trn_mat_time", tmt), ("sequence", seq), ("tst_int", [None] * 299 + [5])):
        with open(d / name, "wb") as f:
            pickle.dump(obj, f)
    args.data = "synth"
    h2 = DataHandler(root=str(tmp_path))
    h2.LoadData()
    assert list(h2.tstUsrs) == [299] and (h2.subMat[0] != tmt[1][0]).nnz == 0
    args.data = "yelp"


def test_powerlaw_generator_properties():
    from sa_gnn_amd import synthetic
    u, i = synthetic.powerlaw_edges(20000, 10000, 200000, seed=1000)
    key = u * 10000 + i
    assert u.numel() <= 200000 and u.numel() > 150000
    assert torch.all(key[1:] > key[:-1])                 # sorted by (user, item), unique
    (rp_u, ci_u), (rp_i, ci_i) = synthetic.csr_pair_from_edges(u, i, 20000, 10000)
    assert rp_u[-1] == u.numel() == rp_i[-1]
    a = sp.csr_matrix((np.ones(u.numel()), ci_u.numpy(), rp_u.numpy()), shape=(20000, 10000))
    b = sp.csr_matrix((np.ones(u.numel()), ci_i.numpy(), rp_i.numpy()), shape=(10000, 20000))
    assert (a - b.T).nnz == 0
    du, di = np.diff(rp_u.numpy()), np.diff(rp_i.numpy())
    assert du.max() > 20 * du.mean() / 4 and di.max() > 30 * di.mean()      # heavy tails
    u2, i2 = synthetic.powerlaw_edges(20000, 10000, 200000, seed=1000)
    assert torch.equal(u, u2) and torch.equal(i, i2)     # deterministic in the seed


def test_write_dataset_roundtrip(tmp_path):
    """The on-disk writer emits what DataHandler.LoadData (the reference's loader contract) reads."""
    from sa_gnn_amd import synthetic
    from sa_gnn_amd.DataHandler import DataHandler
    from sa_gnn_amd.Params import args
    tmt, seq, tst, tdict = synthetic.write_dataset(str(tmp_path / "toy"), 120, 90, [500, 400], test_size=20)
    args.data = "toy"
    h = DataHandler(root=str(tmp_path))
    h.LoadData()
    args.data = "yelp"
    assert (args.user, args.item) == (120, 90) and len(h.subMat) == 2
    assert len(h.tstUsrs) == 60 and h.tstInt[0] == tst[0] and h.tstInt[1] is None
    assert h.test_dict[1] == tdict[1] and min(min(v) for v in h.test_dict.values()) >= 1
    assert h.sequence == seq


def test_sampler_invariants():
    """Vectorised samplers keep the reference's contracts (model.py:252-339): mirrored halves,
    negatives never interacted with nor the last / test item, SSL pairs interleaved and drawn
    from the user's items of that interval."""
    from sa_gnn_amd import synthetic
    from sa_gnn_amd.DataHandler import DataHandler
    from sa_gnn_amd.Params import args
    from sa_gnn_amd.model import Recommender
    args.graphNum, args.batch, args.pos_length, args.sslNum, args.pred_num = 3, 64, 20, 5, 2
    U, I = 300, 200
    tmt = synthetic.make_trn_mat_time(U, I, [3000, 2500, 2000])
    seq = synthetic.make_sequence(tmt)
    tst = [(u * 7) % I if u % 2 else None for u in range(U)]
    h = DataHandler.from_memory(tmt, seq, tst, None)
    rec = Recommender.__new__(Recommender)
    rec.handler = h
    np.random.seed(0)
    bat = np.random.permutation(U)[:64]
    uL, iL, sq, mk, uLs = rec.sampleTrainBatch(bat, h.trnMat, None, 7)
    n = len(uL) // 2
    assert n > 0 and uL[:n] == uL[n:] and uLs[:n] == uLs[n:] and sq.shape == (64, 20)
    for e in range(n):
        u = uL[e]
        assert iL[e] in seq[u][:-1]                                   # positive: an earlier item of the user
        assert h.trnMat[u, iL[n + e]] == 0 and iL[n + e] != seq[u][-1] and iL[n + e] != tst[u]
        assert bat[uLs[e]] == u
    assert set(np.unique(mk)) <= {0.0, 1.0} and np.all((sq != 0) <= (mk != 0) + (sq == 0))
    su, si, sl = rec.sampleSslBatch(bat, h.subMat)
    for k in range(3):
        assert len(su[k]) % 2 == 0 and su[k][0::2] == su[k][1::2]
        for e in range(len(su[k])):
            assert h.subMat[k][su[k][e], si[k][e]] != 0


def test_calc_res_counts_match_the_reference_loop_on_ties_and_duplicates():
    """Recommender.calcRes ranks without sorting (rank = #(pred > p) + #(earlier candidates with pred == p) of the
    best-ranked copy of the target item). Against the oracle's restatement of the reference's sort + list.index loop
    (model.py:484-510) on heavily tied scores and candidate lists that repeat item ids (a pre-drawn negative may
    be the held-out item itself; equal items have equal scores)."""
    from oracle import selfgnn_oracle as O
    from sa_gnn_amd.model import Recommender
    rng = np.random.default_rng(0)
    for _ in range(100):
        B, C = 13, 29
        locs = [rng.integers(0, 12, size=C) for _ in range(B)]
        tem = [int(l[-1]) for l in locs]                                  # the positive is the last candidate
        preds = np.round(rng.standard_normal((B, C)), 1).astype(np.float32)
        for b in range(B):
            for c in range(C):
                preds[b, c] = preds[b, np.argmax(locs[b] == locs[b][c])]
        np.testing.assert_allclose(Recommender.calcRes(preds, tem, locs, shoot=10), O.calc_res(preds, tem, locs, shoot=10))


def test_calc_res_counts_a_nan_score_as_a_miss():
    """A diverged model scores NaN. The reference's sort leaves a NaN positive where it is — last — so it misses;
    the sort-free count must not turn `NaN > x == False` into rank 0 (HR = 1). +-inf keep their meaning."""
    from oracle import selfgnn_oracle as O
    from sa_gnn_amd.model import Recommender
    rng = np.random.default_rng(1)
    B, C = 5, 100
    locs = [list(rng.permutation(1000)[:C]) for _ in range(B)]
    tem = [l[-1] for l in locs]
    preds = rng.standard_normal((B, C))
    preds[0, :] = np.nan             # everything NaN
    preds[1, -1] = np.nan            # only the positive
    preds[2, :50] = np.nan           # half of the negatives: the positive competes with the finite half only
    preds[2, -1] = 10.0
    preds[3, -1] = np.inf
    preds[4, -1] = -np.inf
    got = [Recommender.calcRes(preds[j:j + 1], tem[j:j + 1], locs[j:j + 1], shoot=10) for j in range(B)]
    assert got[0] == (0.0,) * 6 and got[1] == (0.0,) * 6 and got[4] == (0.0,) * 6
    assert got[3] == (1.0,) * 6
    for j in (0, 1, 3, 4):           # where Python's sort is well defined with a NaN in the list
        assert got[j] == O.calc_res(preds[j:j + 1], tem[j:j + 1], locs[j:j + 1], shoot=10)
    assert got[2][0] == 1.0


def test_engine_is_chosen_per_thread_through_the_abi():
    """sagnn_set_engine / sagnn_get_engine: no environment switch, no process-wide state."""
    import threading
    lib = _lib.load()
    assert lib.sagnn_get_engine() == 0
    assert lib.sagnn_set_engine(1) == 0 and lib.sagnn_get_engine() == 1
    seen = []
    th = threading.Thread(target=lambda: seen.append(lib.sagnn_get_engine()))
    th.start(); th.join()
    assert seen == [0]                                   # another thread still runs the default
    assert lib.sagnn_set_engine(7) < 0 and lib.sagnn_get_engine() == 1
    assert lib.sagnn_set_engine(0) == 0


def test_adam_multi_batches_skip_empty_tensors_without_stepping_any_twice():
    """More than 48 tensors (one launch's table) with an empty one among them: host-side argument walk only — the
    launch itself needs a GPU (tests/test_gpu_train.py); here the walk must terminate and reject nothing."""
    import ctypes
    lib = _lib.load()
    assert lib.sagnn_adam_multi_f32(0, None, None, None, None, None, None, 1e-3, 0.9, 0.999, 1e-8, 1, None) == 0



def test_bench_starts_its_own_ranks_when_started_bare(monkeypatch):
    """`python bench.py --gpus N` is how the driver starts every bench (BENCH_rNN.json `cmd`): with N > 1 and no
    WORLD_SIZE in the environment the process must start N ranks under torch.distributed.run as a CHILD (before any
    GPU call), pass its own arguments through and exit with the child's code."""
    import subprocess
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "3", "--warmup", "1"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    seen = {}

    def fake_run(c, env=None, **kw):
        seen["cmd"], seen["env"] = c, env
        return subprocess.CompletedProcess(c, 7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--dist-backend", "gloo", "--scale", "0.004", "--intervals", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                       # the launcher's exit code is the parent's
    assert seen["cmd"][-8:] == ["--gpus", "2", "--dist-backend", "gloo", "--scale", "0.004", "--intervals", "4"]
    assert "--nproc-per-node=2" in seen["cmd"] and seen["env"]["MASTER_ADDR"] == "127.0.0.1"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher (WORLD_SIZE set) a rank count that disagrees with --gpus is still refused
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "agree" in str(e.value.code)
