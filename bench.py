#!/usr/bin/env python3
"""bench.py — SelfGNN interval-propagation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, see launch_command)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one forward pass of the whole hot path over the synthetic power-law workload of
BASELINE.json's roofline configuration (configs[4]): 16 interval graphs of 10M users x 5M items
with ~100M unique edges each, embed_dim 64, 2 GNN layers. The SAME total graph runs at every N
(`--scaling strong`, the default: SURVEY.md §8d defines scaling on the same total graph, and the
16 intervals fit one 288 GB GPU); `--scaling weak` keeps 2 intervals per GPU instead.

    2*T_local*L interval SpMM launches (sagnn_gnn_interval_f32)
    -> exchange of row shards over RCCL (all-to-all)      [N > 1]
    -> interval fusion LSTM -> layer-norm -> MHSA -> mean  (sagnn_interval_fusion_f32)
    -> RCCL all-gather of the fused embeddings            [N > 1]

metric = SpMM edges/s = (edges traversed by all SpMM launches of all ranks per step) / (step time,
max over ranks). `roofline` prices the dominant kernel (spmm_rows_kernel) with HIP events recorded
on its launch stream inside the timed region; `cpu_baseline` times the oracle's C port of the
TF1 CPU op chain on a bounded row sample of the same graph (rank 0, N = 1 only), next to which
the GPU results of the user-side sample, an item-side slice holding the heaviest hub rows and a
>= 100k-row slice of the fused embeddings are compared with the oracle.
Prints exactly one JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: users, items, edges per interval, intervals (strong: total; weak: per GPU), d, L
    "synthetic-powerlaw-10Mx5M": dict(users=10_000_000, items=5_000_000, nnz=100_000_000, t_total=16, t_per_gpu=2, d=64, layers=2),
    # shapes of the real datasets, synthetic edges; all T intervals on every run
    "gowalla-shaped": dict(users=48_653, items=52_619, nnz=600_000, t_total=3, d=64, layers=2),
    "amazon-shaped": dict(users=11_199, items=30_821, nnz=[72280, 78997, 79692, 78096, 45651], t_total=5, d=64, layers=3),
    "movielens-shaped": dict(users=24_312, items=8_681, nnz=300_000, t_total=6, d=128, layers=2),
    # the reference's fourth dataset (yelp.sh:1: graphNum 12, gnn_layer 3; not a BASELINE config); its interaction count is
    # unknown offline — edges per interval of the same order as the other three datasets' totals / T
    "yelp-shaped": dict(users=19_751, items=38_386, nnz=126_000, t_total=12, d=64, layers=3),
}
HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--workload", default="synthetic-powerlaw-10Mx5M", choices=sorted(WORKLOADS))
    p.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                   help="synthetic workload: strong = the 16 intervals of configs[4] at every N; weak = 2 per GPU")
    p.add_argument("--scale", type=float, default=1.0, help="shrink users/items/edges (debug only; recorded in config)")
    p.add_argument("--zipf", type=float, default=0.8,
                   help="item popularity exponent of the generator; 0 = uniform items (the control run: no "
                        "Infinity-Cache-resident hot rows, a plain HBM gather)")
    p.add_argument("--stages", default="full", choices=["full", "spmm", "train"],
                   help="spmm = time the SpMM stack alone; train = forward + backward + Adam of the hot "
                        "path (loss = sum of the fused embeddings; N > 1: reverse all-to-all + reduce-scatter) — "
                        "not the headline metric")
    p.add_argument("--exchange", default="alltoall", choices=["alltoall", "allgather"])
    p.add_argument("--split", default="fractional", choices=["fractional", "groups"],
                   help="T < world: edge-balanced fractional row cuts across all ranks (default) or whole-rank groups per interval")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-breakdown", action="store_true",
                   help="N > 1: skip the extra untimed-for-the-metric passes (SpMM only / exchange only / fusion only)")
    p.add_argument("--cpu-sample-rows", type=int, default=2_000_000)
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work spent on the cpu_baseline sample")
    p.add_argument("--tuning", default="", help="short,long,chunk override for the SpMM plan")
    p.add_argument("--intervals", type=int, default=0, help="override the total interval count (synthetic workload)")
    p.add_argument("--intervals-per-gpu", type=int, default=0, help="weak scaling with this many intervals per GPU")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo = rehearsal of the N>1 pipeline on ONE GPU (every rank uses cuda:0, collectives "
                        "staged through host memory); never a performance run")
    p.add_argument("--dist-single", action="store_true",
                   help="with --gpus 1: initialise the process group (RCCL, or gloo) with ONE rank and run the N > 1 code path — "
                        "exchange rounds, pipelined fusion, gathers, their adjoints — through it; a check of the collective calls, "
                        "never a performance run")
    p.add_argument("--engine", default="f16x2", choices=["f16x2", "f32", "valu"],
                   help="arithmetic engine of the fusion GEMMs for the timed steps (sagnn_set_engine); the default line also "
                        "carries a short same-run block on the exact-fp32 engine (fusion_f32_engine)")
    p.add_argument("--graph", action="store_true",
                   help="N=1: time a hipGraph replay of the step (launch-bound small workloads); the "
                        "per-kernel event timing then comes from an extra eager pass before it")
    return p.parse_args()


def position_checksum(f: torch.Tensor) -> float:
    """sum_r (r + 1) * sum_c |f[r, c]| / (n (n + 1) / 2), float64: changes if rows are permuted or misplaced."""
    n = f.shape[0]
    w = torch.arange(1, n + 1, device=f.device, dtype=torch.float64)
    return float((f.abs().sum(dim=1, dtype=torch.float64) * w).sum() / (n * (n + 1) / 2.0))


def launch_command(n_ranks: int, argv: list, port: int) -> list:
    """The command `python bench.py --gpus N <argv>` runs as a child when it was started bare (no WORLD_SIZE in
    the environment): one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n_ranks: int, argv: list) -> int:
    """Starts the N ranks as a CHILD process (this process has not touched the GPU and never will), relays their
    output — rank 0's one JSON line on stdout — and returns the launcher's exit code."""
    import subprocess
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = launch_command(n_ranks, argv, port)
    log("no WORLD_SIZE in the environment: starting", " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))
    # stdout carries ONE line, the result: whatever the libraries under this process print there (RCCL writes a five-line
    # version banner to stdout when its communicator comes up) is sent to stderr, and the line goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    import torch.distributed as dist
    from sa_gnn_amd import _lib, ops, synthetic
    from sa_gnn_amd.parallel import (ChunkedGather, FractionalRunner, FractionalSharding, RoundFusion, RowShardExchange,
                                     SplitIntervalRunner, SplitIntervalSharding, csr_row_slice, exchange_to_row_shards,
                                     gather_fused, make_sharding)

    # --dist-single: the process group and the N > 1 code path (exchange rounds, pipelined fusion, gathers) with ONE rank —
    # what a one-GPU box can check of the RCCL calls (argument types, devices, contiguity) before an 8-GPU node exists
    multi = world > 1 or a.dist_single
    rehearsal = multi and a.dist_backend == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        if world == 1:                      # --dist-single from a bare shell: a one-rank rendezvous of its own
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if a.dist_single:
        import sa_gnn_amd.parallel as _par
        _par.SINGLE_RANK_COLLECTIVES = True
    lib = _lib.load()
    ops.set_engine(a.engine)

    w = dict(WORKLOADS[a.workload])
    synthetic_wl = "t_per_gpu" in w
    scaling = a.scaling if synthetic_wl else "strong"
    if a.intervals_per_gpu > 0 and synthetic_wl:
        scaling, w["t_per_gpu"] = "weak", a.intervals_per_gpu
    if a.intervals > 0 and synthetic_wl:
        scaling, w["t_total"] = "strong", a.intervals
    U, I = int(w["users"] * a.scale), int(w["items"] * a.scale)
    d, L, heads = w["d"], w["layers"], 16
    T = w["t_per_gpu"] * world if scaling == "weak" else w["t_total"]
    nnz_of = (lambda k: int(w["nnz"][k] * a.scale)) if isinstance(w["nnz"], list) else (lambda k: int(w["nnz"] * a.scale))
    sh = make_sharding(T, world, rank, weights=[nnz_of(k) for k in range(T)], split=a.split)
    frac = isinstance(sh, FractionalSharding)           # T < world: edge-balanced stretches of the concatenated intervals
    split = isinstance(sh, SplitIntervalSharding) or frac     # T < world: ranks split an interval's target rows
    if split and a.exchange != "alltoall":
        raise SystemExit("T < world runs the all-to-all exchange only")
    if split and not frac and a.stages == "train":
        raise SystemExit("--stages train with T < world needs --split fractional (the default): the row-slice stack of the "
                         "whole-rank groups has no backward")
    tuning = tuple(int(v) for v in a.tuning.split(",")) if a.tuning else None
    group = None
    if split:
        groups = [dist.new_group(sh.members(k)) for k in range(T)]     # every rank creates every group, same order
        group = None if frac else groups[sh.interval]

    # ---- build the rank's interval graphs and parameters (untimed) --------------------------
    t0 = time.time()
    plans, emb = [], []
    local_launch_edges = 0                               # edges of one layer's launches (both directions)
    for k in sh.local_intervals:
        u, i = synthetic.powerlaw_edges(U, I, nnz_of(k), seed=1000 + k, device=dev, zipf_s=a.zipf)
        (rp_u, ci_u), (rp_i, ci_i) = synthetic.csr_pair_from_edges(u, i, U, I)
        del u, i
        if split:                                        # this member's target-row slices, full source tables
            (lu, hu), (li, hi) = (sh.slice_range(U, k), sh.slice_range(I, k)) if frac else (sh.slice_range(U), sh.slice_range(I))
            rp_u, ci_u = csr_row_slice(rp_u, ci_u, lu, hu)
            rp_i, ci_i = csr_row_slice(rp_i, ci_i, li, hi)
            pu = ops.SpmmPlan(rp_u.contiguous(), ci_u.contiguous(), hu - lu, I, device=dev, tuning=tuning, validate=False)
            pi = ops.SpmmPlan(rp_i.contiguous(), ci_i.contiguous(), hi - li, U, device=dev, tuning=tuning, validate=False)
        else:
            pu = ops.SpmmPlan(rp_u, ci_u, U, I, device=dev, tuning=tuning, validate=False)
            pi = ops.SpmmPlan(rp_i, ci_i, I, U, device=dev, tuning=tuning, validate=False)
        del rp_u, ci_u, rp_i, ci_i
        g = torch.Generator(device=dev)
        g.manual_seed(2000 + k)
        u0 = (torch.rand((U, d), generator=g, device=dev) * 0.02 - 0.01)
        i0 = (torch.rand((I, d), generator=g, device=dev) * 0.02 - 0.01)
        plans.append((pu, pi))
        emb.append((u0, i0))
        local_launch_edges += pu.nnz + pi.nnz
        log(f"rank {rank}: interval {k}: nnz user-side/item-side={pu.nnz}/{pi.nnz} max_deg={pu.info.max_degree}/{pi.info.max_degree} "
            f"long rows {pu.info.n_long_rows}/{pi.info.n_long_rows} ({time.time() - t0:.1f}s)")
    # dataset-sized graphs: the T intervals of a layer in ONE launch (sagnn_gnn_stack_f32); the embeddings then live in
    # [T, N, d] tensors, as the model's parameters do
    use_batch = (not synthetic_wl) and not multi and len(plans) == T and T > 0
    batch = None
    if use_batch:
        batch = ops.SpmmBatch([pp[0] for pp in plans], [pp[1] for pp in plans])
        emb_u, emb_i = torch.stack([e[0] for e in emb]), torch.stack([e[1] for e in emb])
        emb = [(emb_u[k], emb_i[k]) for k in range(T)]
        scr_bu = torch.empty((2, T, U, d), device=dev) if L > 1 else None
        scr_bi = torch.empty((2, T, I, d), device=dev) if L > 1 else None
    torch.cuda.empty_cache()
    from sa_gnn_amd.model import random_fusion_params
    prm = [random_fusion_params(d, dev, seed) for seed in (7, 8)]       # users, items
    prm[1]["lstm_W"], prm[1]["lstm_b"] = prm[0]["lstm_W"], prm[0]["lstm_b"]     # one shared cell (model.py:141-144)

    t_loc = len(sh.local_intervals)
    overlap = multi and a.exchange == "alltoall"
    comm_dev = torch.device("cpu") if rehearsal else dev          # gloo rehearsal: collectives on host copies
    if frac:
        runner = FractionalRunner(sh, U, I, d, dev, {k: groups[k] for k in sh.local_intervals}, comm_device=comm_dev)
        frac_plans = {k: plans[j] for j, k in enumerate(sh.local_intervals)}
        frac_emb = {k: emb[j] for j, k in enumerate(sh.local_intervals)}
    else:
        runner = SplitIntervalRunner(sh, U, I, d, dev, group=group, comm_device=comm_dev) if split else None
    if not split:
        out_u = torch.empty((max(t_loc, 1), U, d), device=dev)[:t_loc]
        out_i = torch.empty((max(t_loc, 1), I, d), device=dev)[:t_loc]
        scr_u = torch.empty((2, U, d), device=dev) if L > 1 else None
        scr_i = torch.empty((2, I, d), device=dev) if L > 1 else None
    # fusion workspace of the non-pipelined path, sized up front (no allocation inside the timed steps)
    fuse_ws = torch.empty(max(T * max(sh.row_range(U)[1] - sh.row_range(U)[0], sh.row_range(I)[1] - sh.row_range(I)[0]) * d, 1)
                          if a.stages == "full" and not overlap else 1, device=dev)
    state = {}

    # N > 1: row-shard exchange buffers; round j is posted right after interval j's SpMM stack and
    # travels over xGMI under the next interval's SpMMs (--exchange allgather: one blocking
    # all-gather of the stacked outputs instead, for comparison)
    ex_u = RowShardExchange(sh, U, d, comm_dev) if overlap else None
    ex_i = RowShardExchange(sh, I, d, comm_dev) if overlap else None
    pipes = [RoundFusion(ex_u, prm[0], heads, dev), RoundFusion(ex_i, prm[1], heads, dev)] if overlap else None

    def spmm_stack(post: bool):
        if split:
            if frac:
                acc_u, acc_i = runner.run(ops.spmm, frac_plans, frac_emb, L, 0.5)
            else:
                acc_u, acc_i = runner.run(ops.spmm, plans[0][0], plans[0][1], emb[0][0], emb[0][1], L, 0.5)
            if post:
                ex_u.post(acc_u.to(comm_dev))
                ex_i.post(acc_i.to(comm_dev))
            return
        if use_batch:
            ops.gnn_stack(batch, emb_u, emb_i, L, 0.5, out_u, out_i, scr_bu, scr_bi)
            return
        for j in range(t_loc):
            ops.gnn_interval(plans[j][0], plans[j][1], emb[j][0], emb[j][1], L, 0.5, out_u[j], out_i[j], scr_u, scr_i)
            if post:
                ex_u.post(out_u[j].to(comm_dev))
                ex_i.post(out_i[j].to(comm_dev))
        if post:
            # T not a multiple of world: a rank with one interval fewer takes part in the last round with empty sends — HERE, in
            # the order every other rank posts it (users' round, items' round), not lazily when the fusion first waits for the
            # round: by then the other ranks have issued the users' all-gather in between, and collectives issued in different
            # orders on different ranks deadlock (found by tools/fuzz_ranks.py: world 3, T 8)
            for _ in range(t_loc, sh.rounds):
                ex_u.post(None)
                ex_i.post(None)

    def fuse_pipelined():
        # Fusion pipelined with the exchange: the LSTM steps of a round run as soon as that round
        # has arrived (both node types' early rounds first, so they cover the last transfers);
        # the all-gather of the fused users runs under the items' tail, and the items' tail is
        # cut into two row chunks so the first chunk's all-gather runs under the second's compute.
        R = sh.rounds
        for j in range(R - 1):
            for pp in pipes:
                pp.lstm_round(j)
        fins = []
        for idx, (pp, n_rows) in enumerate(zip(pipes, (U, I))):
            rows = pp.ex.rows_local
            if idx == len(pipes) - 1 and n_rows % world == 0 and rows >= 2:
                cg = ChunkedGather(sh, n_rows)
                for lo, hi in ((0, rows // 2), (rows // 2, rows)):
                    pp.lstm_round(R - 1, lo, hi)
                    cg.post(lo, hi, pp.attention(lo, hi).to(comm_dev))
                fins.append(cg.finish)
            else:
                pp.lstm_round(R - 1)
                fins.append(gather_fused(pp.attention().to(comm_dev), sh, n_rows, async_op=True)[1])
            pp.done()
        state["final"] = [fin().to(dev) for fin in fins]

    def step():
        nonlocal fuse_ws
        spmm_stack(post=overlap and a.stages == "full")
        if a.stages == "spmm":
            return
        if overlap:
            fuse_pipelined()
            return
        pending = []
        for x_loc, n_rows, p in ((out_u, U, prm[0]), (out_i, I, prm[1])):
            x = exchange_to_row_shards(x_loc.to(comm_dev), sh, n_rows, mode=a.exchange).to(dev)
            need = x.shape[0] * x.shape[1] * d                                   # [T, rows_local, d]
            if fuse_ws.numel() < need:
                fuse_ws = torch.empty(need, device=dev)
            f = ops.interval_fusion(x.permute(1, 0, 2), p, heads, workspace=fuse_ws)
            pending.append(gather_fused(f.to(comm_dev), sh, n_rows, async_op=True))   # users' gather runs under items' fusion
        state["final"] = [fin().to(dev) for _, fin in pending]

    if a.stages == "train":
        # forward + backward + Adam of the hot path. N > 1: the exchange and the gather carry their adjoints
        # (reverse all-to-all, reduce-scatter: parallel.ExchangeRowsFn / GatherFusedFn), every rank's loss is the
        # sum of ITS rows of the fused embeddings (the total is the N = 1 loss), the replicated fusion weights'
        # gradients are all-reduced, each rank steps its own intervals' embeddings.
        if (split and not frac) or t_loc == 0:
            raise SystemExit("--stages train needs at least one interval (or row stretch) per rank")
        from sa_gnn_amd import autograd as ag
        from sa_gnn_amd.parallel import allreduce_grads, exchange_rows, gather_rows
        if frac:      # T < world: the tables of every interval the rank meets, replicated inside the interval's group
            leaves = {}
            for j, k in enumerate(sh.local_intervals):
                leaves[f"uEmbed{k}"] = emb[j][0].requires_grad_(True)
                leaves[f"iEmbed{k}"] = emb[j][1].requires_grad_(True)
            emb_names = [n_ for n_ in leaves]
        else:
            leaves = {"uEmbed": torch.stack([e[0] for e in emb]).requires_grad_(True),      # [T_local, U, d] / [T_local, I, d]
                      "iEmbed": torch.stack([e[1] for e in emb]).requires_grad_(True)}
            emb_names = ["uEmbed", "iEmbed"]
            emb.clear()
        torch.cuda.empty_cache()
        shared = []
        for tag, p in (("U", prm[0]), ("I", prm[1])):
            for k, v in p.items():
                if tag == "I" and k in ("lstm_W", "lstm_b"):
                    continue                              # the cell is shared: one leaf
                leaves[f"{tag}.{k}"] = v.requires_grad_(True)
                shared.append(v)
        opt = ops.Adam(leaves, lr=1e-3, decay=0.96, decay_step=19, reg=1e-2, reg_names=emb_names)
        if frac:
            from sa_gnn_amd.parallel import FractionalStackFn
        pl_u, pl_i = [pp[0] for pp in plans], [pp[1] for pp in plans]

        def step():                                       # noqa: F811  (training step replaces the forward step)
            for v in leaves.values():
                v.grad = None
            if frac:      # the rank's row slices of every interval it meets; the adjoint all-gathers the table gradients per group
                us, its = FractionalStackFn.apply(runner, ops.spmm_ex, ops.mask_scale, frac_plans, L, 0.5,
                                                  *[leaves[n_] for n_ in emb_names])
            elif use_batch:
                us, its = ag.gnn_stack(leaves["uEmbed"], leaves["iEmbed"], batch, None, L, 0.5)
            else:
                us, its = ag.gnn_stack(leaves["uEmbed"], leaves["iEmbed"], pl_u, pl_i, L, 0.5)  # [T_local, N, d] slabs, no stack copy
            finals, loss = [], 0.0
            for xs, n_rows, p in ((us, U, prm[0]), (its, I, prm[1])):
                if multi:
                    xs = exchange_rows(xs.to(comm_dev), sh, n_rows).to(dev)                        # [T, rows_local, d]
                f_loc = ag.interval_fusion(xs.permute(1, 0, 2), p, heads)
                loss = loss + f_loc.sum()                                                          # this rank's rows
                finals.append(gather_rows(f_loc.to(comm_dev), sh, n_rows).to(dev) if multi else f_loc)
            loss.backward()
            if multi:
                if rehearsal:
                    for v in shared:                      # gloo all-reduces host copies
                        gcpu = v.grad.cpu()
                        dist.all_reduce(gcpu)
                        v.grad.copy_(gcpu)
                else:
                    allreduce_grads(shared)
            opt.step({k: v.grad for k, v in leaves.items()})
            state["final"] = [f.detach() for f in finals]

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    def allmax(x: float) -> float:
        if not multi:
            return x
        tt = torch.tensor([x], device=comm_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    log(f"rank {rank}: setup {time.time() - t0:.1f}s; T={T} ({t_loc} local, {'split' if split else 'whole'} intervals); warmup {a.warmup}")
    for _ in range(a.warmup):
        step()
    sync()
    launches_per_step = max(t_loc, 1) * 2 * L * 2 + 16 if a.stages != "train" else t_loc * 2 * L * 4 + 16
    lib.sagnn_profile_enable(a.steps * launches_per_step + 16)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    ops.range_redo_count(reset=True)
    sync()
    t1 = time.perf_counter()
    marks[0].record()
    for s_ in range(a.steps):
        step()
        marks[s_ + 1].record()                            # per-step device time on the launch stream (median / min)
    sync()
    elapsed = time.perf_counter() - t1
    step_ms = [marks[s_].elapsed_time(marks[s_ + 1]) for s_ in range(a.steps)]
    graph_mode = bool(a.graph and not multi)
    if graph_mode:
        # the eager pass above supplied the HIP-event records; now capture the same step once and
        # time its replays (events are not recorded inside a captured launch sequence)
        cap = a.steps * launches_per_step + 16
        saved = ((ctypes.c_float * cap)(), (ctypes.c_int32 * cap)(), (ctypes.c_int64 * cap)(), (ctypes.c_int64 * cap)(), ctypes.c_int(0))
        _lib.check(lib.sagnn_profile_read(saved[0], saved[1], saved[2], saved[3], cap, ctypes.byref(saved[4])))
        lib.sagnn_profile_enable(0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(a.warmup):
            g.replay()
        sync()
        t1 = time.perf_counter()
        marks[0].record()
        for s_ in range(a.steps):
            g.replay()
            marks[s_ + 1].record()
        sync()
        elapsed = time.perf_counter() - t1
        step_ms = [marks[s_].elapsed_time(marks[s_ + 1]) for s_ in range(a.steps)]
    redo_tiles = ops.range_redo_count()       # tiles the f16 x 2 kernels re-evaluated in fp32 during the timed steps (0 = fast path throughout)
    elapsed = allmax(elapsed)
    if multi:
        te = torch.tensor([local_launch_edges], device=comm_dev, dtype=torch.int64)
        dist.all_reduce(te)
        launch_edges_all = int(te.item())
    else:
        launch_edges_all = local_launch_edges

    # ---- per-launch records from the HIP events -----------------------------------------------
    cap = a.steps * launches_per_step + 16
    if graph_mode:
        ms, kind, ua, ub, n = saved
    else:
        ms = (ctypes.c_float * cap)()
        kind = (ctypes.c_int32 * cap)()
        ua = (ctypes.c_int64 * cap)()
        ub = (ctypes.c_int64 * cap)()
        n = ctypes.c_int(0)
        _lib.check(lib.sagnn_profile_read(ms, kind, ua, ub, cap, ctypes.byref(n)))
        lib.sagnn_profile_enable(0)
    rec = [(kind[i], ms[i], ua[i], ub[i]) for i in range(n.value)]
    rows_k = [r for r in rec if r[0] == 0]
    bytes_per_edge, bytes_per_row = 4 * d + 4, 4 * d + 4 + 4 * d       # residual read is fused

    def price(rs):
        if not rs:
            return None
        b = sum(r[2] * bytes_per_edge + r[3] * bytes_per_row for r in rs)
        t_ms = sum(r[1] for r in rs)
        gbps = b / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
        return {"launches": len(rs), "avg_launch_ms": t_ms / len(rs), "algorithmic_bytes_per_launch": b / len(rs),
                "achieved": gbps, "frac": gbps / HBM_PEAK_GBPS,
                "edges_per_sec": sum(r[2] for r in rs) / (t_ms * 1e-3) if t_ms > 0 else 0.0}

    allk = price(rows_k) or {"launches": 0, "avg_launch_ms": 0.0, "algorithmic_bytes_per_launch": 0.0, "achieved": 0.0, "frac": 0.0}
    # one SpMM call = the row / chunk kernel + (when the graph has long rows) its fix-up launch: the roofline
    # figure prices the CALL; the kernel-only average (what `rocprofv3 --kernel-trace --stats` lists per kernel) is kept beside it
    fix_k = [r for r in rec if r[0] == 1]
    call_ms = (sum(r[1] for r in rows_k) + sum(r[1] for r in fix_k)) / max(len(rows_k), 1)
    kernel_only = {"spmm_rows_kernel_avg_ms": allk["avg_launch_ms"], "spmm_rows_kernel_frac": allk["frac"],
                   "spmm_fixup_kernel_avg_ms": (sum(r[1] for r in fix_k) / len(fix_k)) if fix_k else 0.0,
                   "spmm_fixup_launches": len(fix_k)}
    if rows_k and call_ms > 0:
        allk = dict(allk, avg_launch_ms=call_ms, achieved=allk["algorithmic_bytes_per_launch"] / (call_ms * 1e-3) / 1e9)
        allk["frac"] = allk["achieved"] / HBM_PEAK_GBPS
    n_user_rows = plans[0][0].n_rows if plans else U
    side = {"user_side": price([r for r in rows_k if r[3] == n_user_rows]),        # rows = users, gathers item rows
            "item_side": price([r for r in rows_k if r[3] != n_user_rows])}        # rows = items, gathers user rows
    stage_ms = {name: sum(r[1] for r in rec if r[0] == kk) / a.steps
                for kk, name in ((0, "spmm_rows"), (1, "spmm_fixup"), (2, "lstm"), (3, "layernorm"), (4, "mhsa_mean"))}
    traffic, traffic_source = None, None
    import glob
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic*.json"))):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload") == a.workload and tj.get("scale", 1.0) == a.scale and a.zipf == tj.get("zipf", 0.8):
                traffic = tj.get("bytes_per_launch")
                traffic_source = (f"recorded: profiles/{os.path.basename(tpath)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an "
                                  "earlier run of this workload's SpMM launches, guide corrections applied; mean of the two "
                                  "directions) — NOT measured in this run")
                for k_, v_ in (tj.get("by_direction") or {}).items():
                    if side.get(k_):
                        side[k_]["traffic"] = v_.get("bytes_per_launch")
        except Exception:
            traffic = None

    final_abs_mean = final_pos = None
    if "final" in state:                       # identical on every rank and for every N at equal T
        final_abs_mean = [float(f.double().abs().mean()) for f in state["final"]]
        final_pos = [position_checksum(f) for f in state["final"]]
    edges_per_step = launch_edges_all * L
    value = edges_per_step * a.steps / elapsed
    wl_name = a.workload if a.zipf == 0.8 else f"{a.workload}-zipf{a.zipf:g}"
    result = {
        "metric": "spmm_edges_per_sec", "value": value, "unit": "edges/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl_name, "users": U, "items": I, "intervals_total": T,
                   "intervals_per_gpu": t_loc, "edges_per_interval": launch_edges_all // max(2 * T, 1),
                   "embed_dim": d, "gnn_layers": L, "heads": heads, "stages": a.stages,
                   "exchange": a.exchange if multi else "none", "scale": a.scale, "item_zipf_s": a.zipf,
                   "launch": "hipGraph replay" if graph_mode else "eager",
                   "spmm_launches": ("one per layer for all intervals and both directions (sagnn_gnn_stack_f32)" if use_batch
                                     else "one per interval, layer and direction (sagnn_gnn_interval_f32)"),
                   "fusion_gemm": {"f16x2": "f16x2 (two round-to-nearest f16 pieces, three piece products, fp32 accumulation; "
                                            "the SpMM itself is plain fp32)",
                                   "f32": "f32 MFMA (v_mfma_f32_32x32x2_f32: an fp32 fmaf chain)",
                                   "valu": "VALU fp32"}[ops.get_engine()],
                   "partitioning": ("T < world: rank r computes the stretch [r, r + 1) / world of the intervals' target rows laid end "
                                    f"to end by edge count (intervals met by each rank: {[sh.intervals_of(r) for r in range(world)]}); fusion row-sharded"
                                    if frac else
                                    f"T < world: {sh.group_size} ranks per interval, target rows split inside a group; fusion row-sharded"
                                    if split else f"interval k -> rank k mod {world}; fusion row-sharded")},
        "ms_per_step_rank0": {"median": float(np.median(step_ms)), "min": float(np.min(step_ms)), "max": float(np.max(step_ms)),
                              "note": "device time between step marks on the launch stream; ms_per_step is wall / steps"},
        "roofline": {"bound": "hbm", "kernel": "spmm_rows_kernel", "achieved": allk["achieved"], "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": allk["frac"], "traffic": traffic, "traffic_source": traffic_source,
                     "launches": allk["launches"], "avg_launch_ms": allk["avg_launch_ms"],
                     "avg_launch_note": "per SpMM call: row / chunk kernel + its long-row fix-up launch (kernel_only lists them apart)",
                     "kernel_only": kernel_only,
                     "algorithmic_bytes_per_launch": allk["algorithmic_bytes_per_launch"],
                     "frac_note": "algorithmic bytes / 8 TB/s HBM peak; hot rows served by the 256 MiB Infinity Cache are "
                                  "included, so this is beyond-L2 bandwidth, not pure HBM (a streaming copy reaches 6.3 TB/s)",
                     "by_direction": side},
        "range_redo_tiles_rank0": redo_tiles,
        "stage_ms_per_step_rank0": stage_ms, "final_abs_mean": final_abs_mean, "final_position_checksum": final_pos,
        "spmm_only_edges_per_sec_rank0": (local_launch_edges * L) / ((stage_ms["spmm_rows"] + stage_ms["spmm_fixup"]) * 1e-3)
        if stage_ms["spmm_rows"] > 0 else None,
    }

    # ---- the same steps on the exact-fp32 engine, same run (N = 1): what the f16 x 2 split buys ---------
    if not multi and a.stages == "full" and a.engine == "f16x2" and not graph_mode:
        with ops.engine("f32"):
            step()
            sync()
            tb = time.perf_counter()
            for _ in range(3):
                step()
            sync()
            ms32 = (time.perf_counter() - tb) / 3 * 1e3
        result["fusion_f32_engine"] = {"ms_per_step": ms32, "edges_per_sec": edges_per_step / (ms32 * 1e-3), "steps": 3,
                                       "what": "the same step with sagnn_set_engine(SAGNN_ENGINE_F32): LSTM and attention products on "
                                               "v_mfma_f32_32x32x2_f32 (an fp32 fmaf chain); SpMM unchanged"}
        step()                                                # leave the f16x2 results in place for the checks below
        sync()

    # ---- N > 1: where the step time goes (extra passes, outside the timed region) ---------------
    if multi and a.stages == "full" and not a.no_breakdown:
        bd = {}

        def timed(fn, reps=3):
            fn()
            sync()
            tb = time.perf_counter()
            for _ in range(reps):
                fn()
            sync()
            return allmax((time.perf_counter() - tb) / reps * 1e3)

        try:
            bd["spmm_only"] = timed(lambda: spmm_stack(post=False))
            if overlap:
                def exchange_only():
                    if split:
                        ex_u.post((runner.out_u if frac else runner.acc_u).to(comm_dev))
                        ex_i.post((runner.out_i if frac else runner.acc_i).to(comm_dev))
                    else:
                        for j in range(t_loc):
                            ex_u.post(out_u[j].to(comm_dev))
                            ex_i.post(out_i[j].to(comm_dev))
                    ex_u.finish()
                    ex_i.finish()
                bd["exchange_alltoall_only"] = timed(exchange_only)
                bd["exchange_rounds"] = sh.rounds

                def fusion_only():                          # on the rows already received; no transport
                    for pp in pipes:
                        for j in range(sh.rounds):
                            pp.lstm_round(j, wait=False)
                        pp.attention()
                bd["fusion_only"] = timed(fusion_only)
            lo_u, hi_u = sh.row_range(U)
            fl = torch.empty((hi_u - lo_u, d), device=comm_dev)
            bd["gather_fused_users_only"] = timed(lambda: gather_fused(fl, sh, U))
            need = T * (U + I) * d * 4
            if not split and need < 120e9:                   # the specified all-gather form of the exchange, same data
                def exchange_allgather():
                    for x_loc, n_rows in ((out_u, U), (out_i, I)):
                        exchange_to_row_shards(x_loc.to(comm_dev), sh, n_rows, mode="allgather")
                bd["exchange_allgather_only"] = timed(exchange_allgather, reps=2)
            bd["full_step"] = elapsed / a.steps * 1e3
            bd["note"] = ("ms, max over ranks, each stage alone (no overlap); full_step overlaps exchange rounds with the "
                          "next interval's SpMMs and the gathers with fusion")
        except Exception as e:                               # the metric above is already measured
            bd["error"] = f"{type(e).__name__}: {e}"
        result["breakdown_ms"] = bd

    # ---- CPU baseline + oracle checks on bounded samples (rank 0, N = 1) ------------------------
    if rank == 0 and not multi and not a.no_cpu_baseline and t_loc > 0:
        import scipy.sparse as sp
        from oracle import selfgnn_oracle as O
        from oracle import tf1_path
        pu, pi = plans[0]
        S = min(a.cpu_sample_rows, U)
        rp = pu.rowptr[: S + 1].cpu().numpy()
        ne = int(rp[-1])
        ci = pu.colidx[:ne].cpu().numpy()
        idx = np.empty((ne, 2), dtype=np.int32)
        idx[:, 0] = np.repeat(np.arange(S, dtype=np.int32), np.diff(rp))
        idx[:, 1] = ci
        # the training stage moved the tables into its leaves (and Adam has stepped them: any values serve the check)
        emb0 = (leaves["uEmbed"][0], leaves["iEmbed"][0]) if a.stages == "train" else emb[0]
        src = emb0[1].detach().cpu().numpy()
        threads = tf1_path.max_threads()
        scratch = np.empty((max(ne, 1), d), dtype=np.float32)
        tf1_path.message_propagate(idx, src, S, 0.5, threads=threads, scratch=scratch)      # warm-up
        times = []
        while sum(times) < a.cpu_seconds and len(times) < 200:      # a bounded sample: ~10 s of CPU work
            tc = time.perf_counter()
            cpu_out = tf1_path.message_propagate(idx, src, S, 0.5, threads=threads, scratch=scratch)
            times.append(time.perf_counter() - tc)
        sub = ops.SpmmPlan(rp.copy(), ci.copy(), S, I, device=dev, validate=False)
        gpu_out = ops.spmm(sub, emb0[1].detach(), 0.5).cpu().numpy()
        err = float(np.abs(gpu_out - cpu_out).max())
        # the stronger single-thread CPU point of SURVEY §8d: scipy CSR @ dense, same rows
        A = sp.csr_matrix((np.ones(ne, np.float32), ci, rp), shape=(S, I))
        tc = time.perf_counter()
        sc = A @ src
        t_scipy = time.perf_counter() - tc
        sc = np.maximum(0.5 * sc, sc)
        del A, scratch, idx
        result["cpu_baseline"] = {
            "value": ne / min(times), "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"user-side SpMM of interval {sh.local_intervals[0]}, rows 0..{S - 1} ({ne} edges), "
                      f"gather->segment_sum->leaky as TF1 runs model.py:86-92 on a CPU; best of {len(times)} passes "
                      f"({sum(times):.1f} s of CPU work, mean {ne / (sum(times) / len(times)) / 1e6:.1f} M edges/s)",
            "seconds": min(times), "gpu_vs_cpu_max_abs_err": err,
            "scipy_csr_single_thread_edges_per_sec": ne / t_scipy,
            "scipy_vs_port_max_abs_err": float(np.abs(sc - cpu_out).max())}
        del sc, cpu_out, gpu_out
        # ---- item-side check at FULL size: the heaviest hub rows (chunk + fix-up path) and a block of
        # ordinary rows of the real launch, against the C port of the TF1 op chain
        if a.stages != "train":
            rp_i = pi.rowptr.cpu().numpy().astype(np.int64)
            deg = np.diff(rp_i)
            nb = min(200_000, pi.n_rows)
            hubs = np.argsort(-deg)[:16]
            rows_sel = np.concatenate([np.arange(nb), hubs[hubs >= nb]])       # the block first, then the hubs outside it
            full = ops.spmm(pi, emb[0][0].detach(), 0.5)                     # the full-size item-side launch
            got = full[torch.from_numpy(rows_sel).to(dev)].cpu().numpy()
            del full
            ci_all = pi.colidx
            segs = [ci_all[: int(rp_i[nb])]] + [ci_all[int(rp_i[r]):int(rp_i[r + 1])] for r in rows_sel[nb:]]
            cols = torch.cat(segs).cpu().numpy()
            cnt = deg[rows_sel]
            idx2 = np.empty((int(cnt.sum()), 2), dtype=np.int32)
            idx2[:, 0] = np.repeat(np.arange(len(rows_sel), dtype=np.int32), cnt)
            idx2[:, 1] = cols
            want = tf1_path.message_propagate(idx2, emb[0][0].detach().cpu().numpy(), len(rows_sel), 0.5, threads=threads)
            e_abs = np.abs(got - want)
            # a row's sum carries fp32 rounding proportional to sum_c |x_c| (640k terms on the hubs, added in a
            # different order by the chunked kernel): price the error against that, per row
            sabs = tf1_path.message_propagate(idx2, np.abs(emb[0][0].detach().cpu().numpy()), len(rows_sel), 1.0, threads=threads)
            result["item_side_max_abs_err"] = float(e_abs.max())
            result["item_side_check"] = {"rows": int(len(rows_sel)), "edges": int(cnt.sum()), "max_degree": int(cnt.max()),
                                         "max_abs_ref": float(np.abs(want).max()),
                                         "max_err_over_row_sum_of_abs": float((e_abs / (sabs + 1e-30)).max()),
                                         "worst_over_tolerance": float((e_abs / (1e-5 + 1e-4 * np.abs(want) + 3 * 1.2e-7 * sabs)).max()),
                                         "what": "full-size item-side launch (rows = items, hub rows cut into chunks + fix-up) "
                                                 "vs oracle/c/tf1_path.c on the 16 heaviest rows + the first 200k rows"}
            del got, want, idx2, cols, sabs
        # ---- fused embeddings of >= 100k-row slices of BOTH node types against the numpy oracle (model.py:135-155)
        if a.stages == "full" and "final" in state:
            worst, checks = 0.0, {}
            for tag, xs, n_rows, p, fin in (("users", out_u, U, prm[0], state["final"][0]), ("items", out_i, I, prm[1], state["final"][1])):
                Sf = min(131_072 if tag == "users" else 65_536, n_rows)
                x = xs[:, :Sf, :].permute(1, 0, 2).cpu().numpy()               # [Sf, T, d]
                pnp = {k: v.detach().cpu().numpy() for k, v in p.items()}
                want = O.interval_fusion(np.ascontiguousarray(x), pnp, heads)
                got = fin[:Sf].cpu().numpy()
                e_abs = np.abs(got - want)
                worst = max(worst, float(e_abs.max()))
                checks[tag] = {"rows": int(Sf), "max_abs_err": float(e_abs.max()), "max_abs_ref": float(np.abs(want).max()),
                               "worst_over_tolerance": float((e_abs / (2e-5 + 1e-4 * np.abs(want))).max())}
            result["fused_max_abs_err"] = worst
            result["fused_check"] = {"intervals": int(T), "by_node_type": checks,
                                     "worst_over_tolerance": max(c["worst_over_tolerance"] for c in checks.values()),
                                     "what": "fused embeddings rows 0..S-1 of users and of items of the timed run (LSTM -> LN -> MHSA -> "
                                             "mean) vs oracle/selfgnn_oracle.py interval_fusion on the same propagated rows"}
    if multi:
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    os.close(result_fd)


if __name__ == "__main__":
    main()
