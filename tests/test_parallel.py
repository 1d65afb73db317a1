"""CPU suite, part 3: the interval-sharded exchange (world_size 2 and 3, gloo). Every rank computes
its intervals with the ORACLE (this is a test of the sharding maps and the collectives, not of
the kernels), exchanges, fuses its row shard with the oracle, all-gathers — and must reproduce the
single-process oracle result exactly."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import selfgnn_oracle as O
from sa_gnn_amd.parallel import (ChunkedGather, IntervalSharding, RowShardExchange, exchange_to_row_shards,
                                 gather_fused)


def _problem(T, U, I, d, seed=3):
    rng = np.random.default_rng(seed)
    mats = [sp.csr_matrix((rng.random((U, I)) < 0.1).astype(np.intc)) for _ in range(T)]
    ue = rng.standard_normal((T, U, d)).astype(np.float32)
    ie = rng.standard_normal((T, I, d)).astype(np.float32)
    return mats, ue, ie, O.init_fusion_params(d, rng)


def _worker(rank, world, port, T, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        U, I, d = 23, 17, 16
        mats, ue, ie, p = _problem(T, U, I, d)
        sh = IntervalSharding(T, world, rank)
        outs_u = []
        for k in sh.local_intervals:                     # only this rank's intervals are computed
            u, _ = O.gnn_interval(ue[k], ie[k], O.trans_to_lsts(mats[k])[0],
                                  O.trans_to_lsts(O.transpose(mats[k]))[0], 2, 0.5)
            outs_u.append(u)
        local = torch.from_numpy(np.stack(outs_u, 0)) if outs_u else torch.empty((0, U, d))
        if mode == "incremental":                                    # async rounds, posted one by one
            ex = RowShardExchange(sh, U, d, torch.device("cpu"))
            for j in range(local.shape[0]):
                ex.post(local[j])
            x = ex.finish()
        elif mode == "roundwise":                                    # consume every round as it arrives
            ex = RowShardExchange(sh, U, d, torch.device("cpu"))
            for j in range(local.shape[0]):
                ex.post(local[j])
            slabs = [ex.wait_round(j).clone() for j in range(sh.rounds)]
            assert [s_.shape[0] for s_ in slabs] == [min(world, T - j * world) for j in range(sh.rounds)]
            x = torch.cat(slabs, 0)
            assert torch.equal(ex.finish(), x)
            ex.post(local[0] if local.shape[0] else None)            # the exchange is reusable after finish()
            for j in range(1, local.shape[0]):
                ex.post(local[j])
            assert torch.equal(ex.finish(), x)
        else:
            x = exchange_to_row_shards(local, sh, U, mode=mode)      # [T, rows_local, d]
        lo, hi = sh.row_range(U)
        assert x.shape == (T, hi - lo, d)
        fused = O.interval_fusion(x.permute(1, 0, 2).numpy(), p, 4)
        full = gather_fused(torch.from_numpy(fused), sh, U)
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,T,mode", [(2, 4, "alltoall"), (2, 3, "alltoall"), (3, 4, "alltoall"),
                                          (2, 1, "alltoall"), (2, 3, "allgather"), (2, 3, "incremental"),
                                          (3, 7, "incremental"), (2, 4, "roundwise"), (3, 7, "roundwise"), (2, 1, "roundwise")])
def test_interval_sharded_pipeline_matches_single_process(world, T, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mats, ue, ie, p = _problem(T, 23, 17, 16)
    uv, _ = O.gnn_stack(ue, ie, [O.trans_to_lsts(m)[0] for m in mats],
                        [O.trans_to_lsts(O.transpose(m))[0] for m in mats], 2, 0.5)
    want = O.interval_fusion(uv, p, 4)
    for r in range(world):
        np.testing.assert_allclose(results[r], want, rtol=1e-6, atol=1e-6)


def _two_exchange_worker(rank, world, port, T, q):
    """bench.py's N > 1 forward schedule on CPU tensors: both node types' exchange rounds posted interval by interval, the
    users' all-gather issued while the items' last round is still in flight."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        U, I, d = 24, 18, 8
        rng = np.random.default_rng(11)
        ue = rng.standard_normal((T, U, d)).astype(np.float32)
        ie = rng.standard_normal((T, I, d)).astype(np.float32)
        sh = IntervalSharding(T, world, rank)
        cpu = torch.device("cpu")
        ex_u, ex_i = RowShardExchange(sh, U, d, cpu), RowShardExchange(sh, I, d, cpu)
        for k in sh.local_intervals:
            ex_u.post(torch.from_numpy(ue[k]))
            ex_i.post(torch.from_numpy(ie[k]))
        # a rank with one interval fewer joins the short last round HERE, in the order the others post it. Left to the
        # first wait_round() of each exchange it would come after this rank's users' all-gather, which the other ranks issue
        # AFTER their items' last round: the collectives then meet in different orders and the run hangs.
        for _ in range(len(sh.local_intervals), sh.rounds):
            ex_u.post(None)
            ex_i.post(None)
        fu = torch.cat([ex_u.wait_round(j) for j in range(sh.rounds)], 0).mean(0)      # a stand-in for the fusion of this rank's rows
        full_u, fin_u = gather_fused(fu, sh, U, async_op=True)
        fi = torch.cat([ex_i.wait_round(j) for j in range(sh.rounds)], 0).mean(0)
        full_i = gather_fused(fi, sh, I)
        full_u = fin_u()
        ex_u.finish(), ex_i.finish()
        q.put((rank, full_u.numpy(), full_i.numpy(), ue.mean(0), ie.mean(0)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(3, 8), (2, 3), (3, 4)])
@pytest.mark.timeout(120)
def test_short_last_round_joins_in_collective_order(world, T):
    """T not a multiple of the rank count with TWO exchanges and a gather in between (the schedule of bench.py): every
    rank issues the same sequence of collectives. The GPU rehearsal of (3 ranks, T = 8) hung before the empty rounds were
    posted eagerly (tools/fuzz_ranks.py)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_exchange_worker, args=(r, world, port, T, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p_ in procs:
        p_.join(timeout=30)
        assert p_.exitcode == 0
    for _, fu, fi, want_u, want_i in res:
        np.testing.assert_allclose(fu, want_u, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(fi, want_i, rtol=1e-6, atol=1e-6)


def test_sharding_maps():
    sh = IntervalSharding(16, 8, 3)
    assert sh.local_intervals == [3, 11] and sh.rounds == 2 and sh.owner(11) == 3
    assert IntervalSharding(5, 8, 6).local_intervals == [] and IntervalSharding(5, 8, 4).local_intervals == [4]
    b = IntervalSharding(3, 4, 0).row_bounds(10)
    assert b == [0, 3, 6, 8, 10]
    assert sorted(k for r in range(8) for k in IntervalSharding(13, 8, r).local_intervals) == list(range(13))
    with pytest.raises(ValueError):
        IntervalSharding(4, 2, 2)


def _chunk_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_rows, d = 12 * world, 5
        sh = IntervalSharding(world, world, rank)
        full = torch.arange(n_rows * d, dtype=torch.float32).view(n_rows, d)
        lo_r, hi_r = sh.row_range(n_rows)
        mine = full[lo_r:hi_r]
        cg = ChunkedGather(sh, n_rows)
        for lo, hi in ((0, 5), (5, 12)):                     # uneven chunks of the local rows
            cg.post(lo, hi, mine[lo:hi])
        q.put((rank, torch.equal(cg.finish(), full), torch.equal(gather_fused(mine, sh, n_rows), full)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_chunked_gather_restores_row_order(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p_ in procs:
        p_.join(60)
    assert all(ok1 and ok2 for _, ok1, ok2 in res)


# ---- fewer intervals than ranks: groups of ranks split the target rows of an interval -----------
class _CpuPlan:
    def __init__(self, rowptr, colidx, n_rows, n_src):
        self.n_rows, self.n_src = n_rows, n_src
        self.mat = sp.csr_matrix((np.ones(len(colidx), np.float32), colidx, rowptr), shape=(n_rows, n_src))


def _cpu_spmm(plan, x, leaky, residual=None, out=None, acc_in=None, acc_out=None, want_out=True):
    """Stand-in with ops.spmm's contract (y = leaky(A x) + residual; out = y; acc_out = acc_in + y)."""
    s = torch.from_numpy(plan.mat @ x.numpy())
    y = torch.maximum(leaky * s, s) + residual
    if out is not None:
        out.copy_(y)
    if acc_out is not None:
        acc_out.copy_(acc_in + y)
    return out


def _split_worker(rank, world, port, T, weights, q):
    from sa_gnn_amd.graph import csr_arrays, transpose
    from sa_gnn_amd.parallel import SplitIntervalRunner, SplitIntervalSharding, csr_row_slice, make_sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        U, I, d, L = 23, 17, 16, 3
        mats, ue, ie, p = _problem(T, U, I, d)
        sh = make_sharding(T, world, rank, weights, split="groups")
        assert isinstance(sh, SplitIntervalSharding) and sh.rounds == 1
        groups = [dist.new_group(sh.members(k)) for k in range(T)]      # every rank creates every group
        k = sh.interval
        assert sh.local_intervals == [k]
        rp_u, ci_u = csr_arrays(mats[k])
        rp_i, ci_i = csr_arrays(transpose(mats[k]))
        (lu, hu), (li, hi) = sh.slice_range(U), sh.slice_range(I)
        pu = _CpuPlan(*csr_row_slice(rp_u, ci_u, lu, hu), hu - lu, I)
        pi = _CpuPlan(*csr_row_slice(rp_i, ci_i, li, hi), hi - li, U)
        run = SplitIntervalRunner(sh, U, I, d, torch.device("cpu"), group=groups[k])
        acc_u, acc_i = run.run(_cpu_spmm, pu, pi, torch.from_numpy(ue[k]), torch.from_numpy(ie[k]), L, 0.5)
        res = {}
        for tag, acc, n_rows in (("u", acc_u, U), ("i", acc_i, I)):
            ex = RowShardExchange(sh, n_rows, d, torch.device("cpu"))
            ex.post(acc)
            x = ex.wait_round(0).clone()
            assert torch.equal(ex.finish(), x)
            assert torch.equal(exchange_to_row_shards(acc, sh, n_rows), x)      # blocking form, same splits
            lo, hi_ = sh.row_range(n_rows)
            assert x.shape == (T, hi_ - lo, d)
            fused = O.interval_fusion(x.permute(1, 0, 2).numpy(), p, 4) if hi_ > lo else np.zeros((0, d), np.float32)
            res[tag] = gather_fused(torch.from_numpy(fused), sh, n_rows).numpy()
        q.put((rank, res["u"], res["i"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T,weights", [(8, 5, [72280, 78997, 79692, 78096, 45651]), (4, 3, None), (3, 1, None),
                                             (8, 3, [5, 1, 1])])
def test_split_interval_groups_match_single_process(world, T, weights):
    """T < world (Amazon T = 5 / Gowalla T = 3 on 8 GPUs): rank groups split the target rows of an
    interval, all-gather the layer outputs inside the group, and feed ONE all-to-all; the fused
    embeddings of both node types must equal the single-process oracle on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, T, weights, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mats, ue, ie, p = _problem(T, 23, 17, 16)
    uv, iv = O.gnn_stack(ue, ie, [O.trans_to_lsts(m)[0] for m in mats],
                         [O.trans_to_lsts(O.transpose(m))[0] for m in mats], 3, 0.5)
    want_u, want_i = O.interval_fusion(uv, p, 4), O.interval_fusion(iv, p, 4)
    for _, got_u, got_i in results:
        np.testing.assert_allclose(got_u, want_u, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(got_i, want_i, rtol=1e-5, atol=1e-5)


def test_split_sharding_maps():
    from sa_gnn_amd.parallel import SplitIntervalSharding, make_sharding
    w = [72280, 78997, 79692, 78096, 45651]
    shs = [SplitIntervalSharding(5, 8, r, w) for r in range(8)]
    assert shs[0].group_size == [1, 2, 2, 2, 1]                      # the three heaviest intervals get two ranks
    assert [s.interval for s in shs] == [0, 1, 1, 2, 2, 3, 3, 4] and [s.member for s in shs] == [0, 0, 1, 0, 1, 0, 1, 0]
    for n_rows in (11199, 30821, 7, 1):
        for k in range(5):                                           # member slices tile [0, n_rows)
            cuts = [shs[r].slice_range(n_rows) for r in shs[0].members(k)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n_rows and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        tot_in = sum(sum(s.exchange_splits(n_rows)[0]) for s in shs)
        tot_out = sum(sum(s.exchange_splits(n_rows)[1]) for s in shs)
        assert tot_in == tot_out == 5 * n_rows
        for r in range(8):                                           # what r receives from s = what s sends to r
            assert shs[r].exchange_splits(n_rows)[1] == [shs[s].exchange_splits(n_rows)[0][r] for s in range(8)]
    assert type(make_sharding(16, 8, 0)) is IntervalSharding and type(make_sharding(8, 8, 0)) is IntervalSharding
    assert SplitIntervalSharding(3, 8, 0, None).group_size == [3, 3, 2]
    with pytest.raises(ValueError):
        SplitIntervalSharding(8, 8, 0)


# ---- backward of the exchange: reduce-scatter / reverse all-to-all ------------------------------------
def _bwd_worker(rank, world, port, T, q):
    from sa_gnn_amd.parallel import allreduce_grads, exchange_rows, gather_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        U, d = 23, 16
        rng = np.random.default_rng(11)
        leaves_all = rng.standard_normal((T, U, d))                    # one "interval output" per interval
        wts = rng.standard_normal((world, U, d))                       # rank r's loss weights (its own batch)
        p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in O.init_fusion_params(d, rng).items()}
        sh = IntervalSharding(T, world, rank)
        mine = sh.local_intervals
        leaf = torch.tensor(leaves_all[mine] if mine else np.zeros((0, U, d)), dtype=torch.float64, requires_grad=True)
        x = exchange_rows(torch.tanh(leaf), sh, U)                     # [T, rows_local, d]
        f_loc = O.torch_interval_fusion(x.permute(1, 0, 2), p, 4)      # [rows_local, d]
        F = gather_rows(f_loc, sh, U)                                  # [U, d] everywhere
        loss = (F * torch.tensor(wts[rank])).sum()                     # every rank a DIFFERENT loss term
        loss.backward()
        allreduce_grads(p.values())
        q.put((rank, mine, leaf.grad.numpy(), {k: v.grad.numpy() for k, v in p.items()}, float(loss)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(2, 4), (3, 7), (2, 3), (3, 2)])
def test_distributed_backward_matches_single_process(world, T):
    """Gradients through gather (adjoint: reduce-scatter) and exchange (adjoint: reverse all-to-all), with a
    different loss term on every rank, against single-process float64 autograd of the summed loss."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bwd_worker, args=(r, world, port, T, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    U, d = 23, 16
    rng = np.random.default_rng(11)
    leaves_all = rng.standard_normal((T, U, d))
    wts = rng.standard_normal((world, U, d))
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in O.init_fusion_params(d, rng).items()}
    leaf = torch.tensor(leaves_all, dtype=torch.float64, requires_grad=True)
    F = O.torch_interval_fusion(torch.tanh(leaf).permute(1, 0, 2), p, 4)
    (F * torch.tensor(wts.sum(0))).sum().backward()
    assert abs(sum(r[4] for r in res) - float((F * torch.tensor(wts.sum(0))).sum())) < 1e-9
    for rank, mine, g_leaf, g_p, _ in res:
        if mine:
            np.testing.assert_allclose(g_leaf, leaf.grad.numpy()[mine], rtol=1e-9, atol=1e-12)
        for k in p:
            np.testing.assert_allclose(g_p[k], p[k].grad.numpy(), rtol=1e-9, atol=1e-11)


# ---- fewer intervals than ranks, edge-balanced: a rank takes the tail rows of one interval and the head rows of the next
def _fractional_worker(rank, world, port, T, weights, q):
    from sa_gnn_amd.graph import csr_arrays, transpose
    from sa_gnn_amd.parallel import FractionalRunner, FractionalSharding, csr_row_slice, make_sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        U, I, d, L = 23, 17, 16, 3
        mats, ue, ie, p = _problem(T, U, I, d)
        sh = make_sharding(T, world, rank, weights)
        assert isinstance(sh, FractionalSharding) and sh.rounds == 1
        groups = [dist.new_group(sh.members(k)) for k in range(T)]      # every rank creates every group
        plans, emb = {}, {}
        for k in sh.intervals_of(rank):
            rp_u, ci_u = csr_arrays(mats[k])
            rp_i, ci_i = csr_arrays(transpose(mats[k]))
            (lu, hu), (li, hi) = sh.slice_range(U, k), sh.slice_range(I, k)
            plans[k] = (_CpuPlan(*csr_row_slice(rp_u, ci_u, lu, hu), hu - lu, I), _CpuPlan(*csr_row_slice(rp_i, ci_i, li, hi), hi - li, U))
            emb[k] = (torch.from_numpy(ue[k]), torch.from_numpy(ie[k]))
        run = FractionalRunner(sh, U, I, d, torch.device("cpu"), {k: groups[k] for k in sh.intervals_of(rank)})
        acc_u, acc_i = run.run(_cpu_spmm, plans, emb, L, 0.5)
        res = {}
        for tag, acc, n_rows in (("u", acc_u, U), ("i", acc_i, I)):
            ex = RowShardExchange(sh, n_rows, d, torch.device("cpu"))
            ex.post(acc)
            x = ex.wait_round(0).clone()
            assert torch.equal(ex.finish(), x)
            assert torch.equal(exchange_to_row_shards(acc, sh, n_rows), x)      # blocking form, same splits
            lo, hi_ = sh.row_range(n_rows)
            assert x.shape == (T, hi_ - lo, d)
            fused = O.interval_fusion(x.permute(1, 0, 2).numpy(), p, 4) if hi_ > lo else np.zeros((0, d), np.float32)
            res[tag] = gather_fused(torch.from_numpy(fused), sh, n_rows).numpy()
        q.put((rank, res["u"], res["i"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T,weights", [(8, 5, [72280, 78997, 79692, 78096, 45651]), (4, 3, None), (3, 1, None),
                                             (8, 3, [5, 1, 1]), (6, 4, [1, 30, 1, 1])])
def test_fractional_sharding_matches_single_process(world, T, weights):
    """T < world, edge-balanced (Amazon's T = 5 on 8 ranks: 44 k edges on every rank instead of 72 k on the busiest):
    ranks take fractional stretches of the concatenated intervals, all-gather layer outputs inside every interval's
    group (uneven slices, a rank in two groups), and feed ONE all-to-all; the fused embeddings of both node types
    equal the single-process oracle on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fractional_worker, args=(r, world, port, T, weights, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mats, ue, ie, p = _problem(T, 23, 17, 16)
    uv, iv = O.gnn_stack(ue, ie, [O.trans_to_lsts(m)[0] for m in mats],
                         [O.trans_to_lsts(O.transpose(m))[0] for m in mats], 3, 0.5)
    want_u, want_i = O.interval_fusion(uv, p, 4), O.interval_fusion(iv, p, 4)
    for _, got_u, got_i in results:
        np.testing.assert_allclose(got_u, want_u, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(got_i, want_i, rtol=1e-5, atol=1e-5)


def test_fractional_sharding_maps_balance_edges():
    from sa_gnn_amd.parallel import FractionalSharding, SplitIntervalSharding
    w = [72280, 78997, 79692, 78096, 45651]
    shs = [FractionalSharding(5, 8, r, w) for r in range(8)]
    assert [s.intervals_of(s.rank) for s in shs] == [[0], [0, 1], [1], [1, 2], [2], [2, 3], [3, 4], [4]]
    assert shs[0].members(1) == [1, 2, 3] and shs[0].members(4) == [6, 7]
    for n_rows in (11199, 30821, 7, 1):
        for k in range(5):                                           # member slices tile [0, n_rows)
            c = shs[0].cuts(n_rows, k)
            assert c[0] == 0 and c[-1] == n_rows and all(a <= b for a, b in zip(c, c[1:]))
            assert [shs[r].slice_range(n_rows, k) for r in shs[0].members(k)] == list(zip(c, c[1:]))
        assert sum(sum(s.exchange_splits(n_rows)[0]) for s in shs) == 5 * n_rows == sum(sum(s.exchange_splits(n_rows)[1]) for s in shs)
        for s in shs:                                                # the send permutation covers the rank's rows once
            pieces = s.send_order(n_rows)
            tot = sum(hi - lo for lo, hi in (s.slice_range(n_rows, k) for k in s.intervals_of(s.rank)))
            assert sorted(i for o, n in pieces for i in range(o, o + n)) == list(range(tot))
    # edges per rank (rows of an interval taken as equally heavy): 44.3 k on every rank; whole-rank groups: 72 k on the busiest
    U = 11199
    load = [sum((s.slice_range(U, k)[1] - s.slice_range(U, k)[0]) / U * w[k] for k in s.intervals_of(s.rank)) for s in shs]
    assert max(load) <= 1.001 * sum(w) / 8
    grp = [SplitIntervalSharding(5, 8, r, w) for r in range(8)]
    load_g = [(g.slice_range(U)[1] - g.slice_range(U)[0]) / U * w[g.interval] for g in grp]
    assert max(load_g) >= 1.6 * sum(w) / 8


# ---- training with fewer intervals than ranks: the row-slice stack and the one all-to-all carry their adjoints
def _bits(slopes_one):
    """[rows, d] bool (True = slope 1) -> [rows, d/4] uint8, bit j of byte l = column 4 l + j (sagnn_spmm_ex_f32's layout)."""
    b = slopes_one.reshape(slopes_one.shape[0], -1, 4).to(torch.uint8)
    return (b[..., 0] | (b[..., 1] << 1) | (b[..., 2] << 2) | (b[..., 3] << 3)).contiguous()


def _unbits(mask, d):
    m = mask.to(torch.int32)
    return torch.stack([(m >> j) & 1 for j in range(4)], dim=-1).reshape(mask.shape[0], d).bool()


def _cpu_spmm_ex(plan, x, leaky, residual=None, out=None, acc_in=None, acc_out=None, want_out=True, acc_in2=None,
                 mask_out=None, mask_in=None, out2=None, slope2=1.0):
    """ops.spmm_ex's contract on the CPU (sagnn_spmm_ex_f32's epilogue, finish_row in csrc/spmm.hip)."""
    s = torch.from_numpy(plan.mat @ x.numpy())
    y = torch.maximum(leaky * s, s)
    if mask_out is not None:
        mask_out.copy_(_bits(s > leaky * s))
    if residual is not None:
        y = y + residual
    if out is not None:
        out.copy_(y)
    v = y
    if acc_out is not None:
        v = y + (acc_in if acc_in is not None else 0) + (acc_in2 if acc_in2 is not None else 0)
        acc_out.copy_(v)
    if out2 is not None:
        keep = _unbits(mask_in, v.shape[1]) if mask_in is not None else torch.ones_like(v, dtype=torch.bool)
        out2.copy_(torch.where(keep, v, slope2 * v))
    return out


def _cpu_mask_scale(g, mask, slope, out):
    out.copy_(torch.where(_unbits(mask, g.shape[1]), g, slope * g))
    return out


def _frac_train_worker(rank, world, port, T, weights, q):
    from sa_gnn_amd.graph import csr_arrays, transpose
    from sa_gnn_amd.parallel import FractionalRunner, FractionalStackFn, csr_row_slice, exchange_rows, make_sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        U, I, d, L = 23, 17, 16, 3
        mats, ue, ie, _ = _problem(T, U, I, d)
        rng = np.random.default_rng(5)
        w_u, w_i = rng.standard_normal((T, U, d)).astype(np.float32), rng.standard_normal((T, I, d)).astype(np.float32)
        sh = make_sharding(T, world, rank, weights)
        groups = [dist.new_group(sh.members(k)) for k in range(T)]
        plans, leaves = {}, []
        for k in sh.intervals_of(rank):
            rp_u, ci_u = csr_arrays(mats[k])
            rp_i, ci_i = csr_arrays(transpose(mats[k]))
            (lu, hu), (li, hi) = sh.slice_range(U, k), sh.slice_range(I, k)
            plans[k] = (_CpuPlan(*csr_row_slice(rp_u, ci_u, lu, hu), hu - lu, I), _CpuPlan(*csr_row_slice(rp_i, ci_i, li, hi), hi - li, U))
            leaves += [torch.from_numpy(ue[k]).requires_grad_(True), torch.from_numpy(ie[k]).requires_grad_(True)]
        run = FractionalRunner(sh, U, I, d, torch.device("cpu"), {k: groups[k] for k in sh.intervals_of(rank)})
        ou, oi = FractionalStackFn.apply(run, _cpu_spmm_ex, _cpu_mask_scale, plans, L, 0.5, *leaves)
        loss = 0.0
        for o_, n_rows, w in ((ou, U, w_u), (oi, I, w_i)):
            x = exchange_rows(o_, sh, n_rows)                               # [T, rows_local, d]
            lo, hi_ = sh.row_range(n_rows)
            loss = loss + (x * torch.from_numpy(w[:, lo:hi_])).sum()        # this rank's rows of the total loss
        loss.backward()
        q.put((rank, sh.intervals_of(rank), [t.grad.numpy() for t in leaves], float(loss)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T,weights", [(8, 5, [72280, 78997, 79692, 78096, 45651]), (4, 3, None), (3, 2, [3, 1])])
def test_fractional_training_gradients_match_single_process(world, T, weights):
    """Forward + backward of the T < world pipeline up to the exchange: the row-slice stack records its activation masks,
    the backward runs the same row-slice SpMMs on all-gathered masked gradient tables, the all-to-all runs in reverse
    (undoing the send permutation of two-segment ranks), and every member of an interval's group ends with the WHOLE
    gradient of the interval's embedding tables. Against float64 autograd of the single-process loss."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frac_train_worker, args=(r, world, port, T, weights, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    U, I, d, L = 23, 17, 16, 3
    mats, ue, ie, _ = _problem(T, U, I, d)
    rng = np.random.default_rng(5)
    w_u, w_i = rng.standard_normal((T, U, d)).astype(np.float32), rng.standard_normal((T, I, d)).astype(np.float32)
    tu = torch.tensor(ue, dtype=torch.float64, requires_grad=True)
    ti = torch.tensor(ie, dtype=torch.float64, requires_grad=True)
    total = 0.0
    for k in range(T):
        a, b = O.torch_gnn_interval(tu[k], ti[k], O.trans_to_lsts(mats[k])[0], O.trans_to_lsts(O.transpose(mats[k]))[0], L, 0.5)
        total = total + (a * torch.tensor(w_u[k], dtype=torch.float64)).sum() + (b * torch.tensor(w_i[k], dtype=torch.float64)).sum()
    total.backward()
    assert abs(sum(r[3] for r in res) - float(total)) <= 1e-4 * abs(float(total)) + 1e-3
    for rank, ks, grads, _ in res:
        for j, k in enumerate(ks):
            np.testing.assert_allclose(grads[2 * j], tu.grad[k].numpy(), rtol=2e-4, atol=2e-4)
            np.testing.assert_allclose(grads[2 * j + 1], ti.grad[k].numpy(), rtol=2e-4, atol=2e-4)
