#!/usr/bin/env python3
"""bench.py — SelfGNN interval-propagation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one forward pass of the whole hot path over the synthetic power-law workload of
BASELINE.json's roofline configuration, weak-scaled: every GPU owns 2 interval graphs of
10M users x 5M items with ~100M unique edges each (so N = 8 is the quoted 16-interval run):

    2*T_local*L interval SpMM launches (sagnn_gnn_interval_f32)
    -> exchange of row shards over RCCL (all-to-all)      [N > 1]
    -> interval fusion LSTM -> layer-norm -> MHSA -> mean  (sagnn_interval_fusion_f32)
    -> RCCL all-gather of the fused embeddings            [N > 1]

metric = SpMM edges/s = (edges traversed by all SpMM launches of all ranks per step) / (step time,
max over ranks). `roofline` prices the dominant kernel (spmm_rows_kernel) with HIP events recorded
on its launch stream inside the timed region; `cpu_baseline` times the oracle's C port of the
TF1 CPU op chain on a bounded row sample of the same graph (rank 0, N = 1 only).
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
    # name: users, items, edges per interval, intervals per GPU, d, L, heads
    "synthetic-powerlaw-10Mx5M": dict(users=10_000_000, items=5_000_000, nnz=100_000_000, t_per_gpu=2, d=64, layers=2),
    # shapes of the real datasets, synthetic edges; all T intervals on every run
    "gowalla-shaped": dict(users=48_653, items=52_619, nnz=600_000, t_total=3, d=64, layers=2),
    "amazon-shaped": dict(users=11_199, items=30_821, nnz=[72280, 78997, 79692, 78096, 45651], t_total=5, d=64, layers=3),
    "movielens-shaped": dict(users=24_312, items=8_681, nnz=300_000, t_total=6, d=128, layers=2),
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
    p.add_argument("--scale", type=float, default=1.0, help="shrink users/items/edges (debug only; recorded in config)")
    p.add_argument("--stages", default="full", choices=["full", "spmm", "train"],
                   help="spmm = time the SpMM stack alone; train = forward + backward + Adam of the hot "
                        "path (N = 1; loss = sum of the fused embeddings) — not the headline metric")
    p.add_argument("--exchange", default="alltoall", choices=["alltoall", "allgather"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-rows", type=int, default=2_000_000)
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work spent on the cpu_baseline sample")
    p.add_argument("--tuning", default="", help="short,long,chunk override for the SpMM plan")
    p.add_argument("--intervals-per-gpu", type=int, default=0, help="override (synthetic workload only)")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo = rehearsal of the N>1 pipeline on ONE GPU (every rank uses cuda:0, collectives "
                        "staged through host memory); never a performance run")
    p.add_argument("--graph", action="store_true",
                   help="N=1: time a hipGraph replay of the step (launch-bound small workloads); the "
                        "per-kernel event timing then comes from an extra eager pass before it")
    return p.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    import torch.distributed as dist
    from sa_gnn_amd import _lib, ops, synthetic
    from sa_gnn_amd.parallel import (IntervalSharding, RoundFusion, RowShardExchange, exchange_to_row_shards,
                                     ChunkedGather, gather_fused)

    rehearsal = world > 1 and a.dist_backend == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    lib = _lib.load()

    w = dict(WORKLOADS[a.workload])
    if a.intervals_per_gpu > 0 and "t_per_gpu" in w:
        w["t_per_gpu"] = a.intervals_per_gpu
    U, I = int(w["users"] * a.scale), int(w["items"] * a.scale)
    d, L, heads = w["d"], w["layers"], 16
    T = w["t_per_gpu"] * world if "t_per_gpu" in w else w["t_total"]
    nnz_of = (lambda k: int(w["nnz"][k] * a.scale)) if isinstance(w["nnz"], list) else (lambda k: int(w["nnz"] * a.scale))
    sh = IntervalSharding(T, world, rank)
    tuning = tuple(int(v) for v in a.tuning.split(",")) if a.tuning else None

    # ---- build the rank's interval graphs and parameters (untimed) --------------------------
    t0 = time.time()
    plans, emb = [], []
    local_edges = 0
    for k in sh.local_intervals:
        u, i = synthetic.powerlaw_edges(U, I, nnz_of(k), seed=1000 + k, device=dev)
        (rp_u, ci_u), (rp_i, ci_i) = synthetic.csr_pair_from_edges(u, i, U, I)
        del u, i
        pu = ops.SpmmPlan(rp_u, ci_u, U, I, device=dev, tuning=tuning, validate=False)
        pi = ops.SpmmPlan(rp_i, ci_i, I, U, device=dev, tuning=tuning, validate=False)
        g = torch.Generator(device=dev)
        g.manual_seed(2000 + k)
        u0 = (torch.rand((U, d), generator=g, device=dev) * 0.02 - 0.01)
        i0 = (torch.rand((I, d), generator=g, device=dev) * 0.02 - 0.01)
        plans.append((pu, pi))
        emb.append((u0, i0))
        local_edges += pu.nnz
        log(f"rank {rank}: interval {k}: nnz={pu.nnz} max_deg user/item={pu.info.max_degree}/{pi.info.max_degree} "
            f"long rows {pu.info.n_long_rows}/{pi.info.n_long_rows} ({time.time() - t0:.1f}s)")
    torch.cuda.empty_cache()
    from sa_gnn_amd.model import random_fusion_params
    prm = [random_fusion_params(d, dev, seed) for seed in (7, 8)]       # users, items
    prm[1]["lstm_W"], prm[1]["lstm_b"] = prm[0]["lstm_W"], prm[0]["lstm_b"]     # one shared cell (model.py:141-144)

    t_loc = len(sh.local_intervals)
    out_u = torch.empty((max(t_loc, 1), U, d), device=dev)[:t_loc]
    out_i = torch.empty((max(t_loc, 1), I, d), device=dev)[:t_loc]
    scr_u = torch.empty((2, U, d), device=dev) if L > 1 else None
    scr_i = torch.empty((2, I, d), device=dev) if L > 1 else None
    # fusion workspace of the non-pipelined path, sized up front (no allocation inside the timed steps)
    fuse_ws = torch.empty(max(T * max(sh.row_range(U)[1] - sh.row_range(U)[0], sh.row_range(I)[1] - sh.row_range(I)[0]) * d, 1)
                          if a.stages == "full" and not (world > 1 and a.exchange == "alltoall") else 1, device=dev)
    state = {}

    # N > 1: row-shard exchange buffers; round j is posted right after interval j's SpMM stack and
    # travels over xGMI under the next interval's SpMMs (--exchange allgather: one blocking
    # all-gather of the stacked outputs instead, for comparison)
    overlap = world > 1 and a.exchange == "alltoall"
    comm_dev = torch.device("cpu") if rehearsal else dev          # gloo rehearsal: collectives on host copies
    ex_u = RowShardExchange(sh, U, d, comm_dev) if overlap else None
    ex_i = RowShardExchange(sh, I, d, comm_dev) if overlap else None
    pipes = [RoundFusion(ex_u, prm[0], heads, dev), RoundFusion(ex_i, prm[1], heads, dev)] if overlap else None

    def step():
        nonlocal fuse_ws
        for j in range(t_loc):
            ops.gnn_interval(plans[j][0], plans[j][1], emb[j][0], emb[j][1], L, 0.5, out_u[j], out_i[j], scr_u, scr_i)
            if overlap and a.stages == "full":
                ex_u.post(out_u[j].to(comm_dev))
                ex_i.post(out_i[j].to(comm_dev))
        if a.stages == "spmm":
            return
        if overlap:
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
        if world != 1:
            raise SystemExit("--stages train is a single-GPU measurement")
        from sa_gnn_amd import autograd as ag
        leaves = {}
        for j in range(t_loc):
            leaves[f"u{j}"] = emb[j][0].requires_grad_(True)
            leaves[f"i{j}"] = emb[j][1].requires_grad_(True)
        for tag, p in (("U", prm[0]), ("I", prm[1])):
            for k, v in p.items():
                if tag == "I" and k in ("lstm_W", "lstm_b"):
                    continue                              # the cell is shared: one leaf
                leaves[f"{tag}.{k}"] = v.requires_grad_(True)
        opt = ops.Adam(leaves, lr=1e-3, decay=0.96, decay_step=19, reg=1e-2,
                       reg_names=[k for k in leaves if k[0] in "ui"])

        def step():                                       # noqa: F811  (training step replaces the forward step)
            for v in leaves.values():
                v.grad = None
            us, its = [], []
            for j in range(t_loc):
                uo, io = ag.gnn_interval(leaves[f"u{j}"], leaves[f"i{j}"], plans[j][0], plans[j][1], L, 0.5)
                us.append(uo)
                its.append(io)
            fu = ag.interval_fusion(torch.stack(us).permute(1, 0, 2), prm[0], heads)
            fi = ag.interval_fusion(torch.stack(its).permute(1, 0, 2), prm[1], heads)
            (fu.sum() + fi.sum()).backward()
            opt.step({k: v.grad for k, v in leaves.items()})
            state["final"] = [fu.detach(), fi.detach()]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    log(f"rank {rank}: setup {time.time() - t0:.1f}s; warmup {a.warmup}")
    for _ in range(a.warmup):
        step()
    sync()
    launches_per_step = t_loc * 2 * L * 2 + 8 if a.stages != "train" else t_loc * 2 * L * 4 + 16
    lib.sagnn_profile_enable(a.steps * launches_per_step + 16)
    sync()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t1
    graph_mode = bool(a.graph and world == 1)
    if graph_mode:
        # the eager pass above supplied the HIP-event records; now capture the same step once and
        # time its replays (events are not recorded inside a captured launch sequence)
        cap = a.steps * launches_per_step + 16
        _ms, _n = (ctypes.c_float * cap)(), ctypes.c_int(0)
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
        for _ in range(a.steps):
            g.replay()
        sync()
        elapsed = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([elapsed], device=comm_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        te = torch.tensor([local_edges], device=comm_dev, dtype=torch.int64)
        dist.all_reduce(te)
        total_edges_once = int(te.item())
    else:
        total_edges_once = local_edges

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
    alg_bytes = sum(r[2] * bytes_per_edge + r[3] * bytes_per_row for r in rows_k)
    k_ms = sum(r[1] for r in rows_k)
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    stage_ms = {name: sum(r[1] for r in rec if r[0] == kk) / a.steps
                for kk, name in ((0, "spmm_rows"), (1, "spmm_fixup"), (2, "lstm"), (3, "layernorm"), (4, "mhsa_mean"))}
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload") == a.workload and tj.get("scale", 1.0) == a.scale:
                traffic = tj.get("bytes_per_launch")
        except Exception:
            traffic = None

    checksum = None
    if "final" in state:                       # identical on every rank and for every N at equal T
        checksum = [float(f.double().abs().mean()) for f in state["final"]]
    edges_per_step = total_edges_once * 2 * L
    value = edges_per_step * a.steps / elapsed
    result = {
        "metric": "spmm_edges_per_sec", "value": value, "unit": "edges/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": a.workload, "users": U, "items": I, "intervals_total": T,
                   "intervals_per_gpu": t_loc, "edges_per_interval": total_edges_once // max(T, 1),
                   "embed_dim": d, "gnn_layers": L, "heads": heads, "stages": a.stages,
                   "exchange": a.exchange if world > 1 else "none", "scale": a.scale,
                   "launch": "hipGraph replay" if graph_mode else "eager",
                   "partitioning": f"interval k -> rank k mod {world}; fusion row-sharded"},
        "roofline": {"bound": "hbm", "kernel": "spmm_rows_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "launches": len(rows_k), "avg_launch_ms": k_ms / max(len(rows_k), 1),
                     "algorithmic_bytes_per_launch": alg_bytes / max(len(rows_k), 1)},
        "stage_ms_per_step_rank0": stage_ms, "final_abs_mean": checksum,
        "spmm_only_edges_per_sec_rank0": (local_edges * 2 * L) / ((stage_ms["spmm_rows"] + stage_ms["spmm_fixup"]) * 1e-3)
        if stage_ms["spmm_rows"] > 0 else None,
    }

    # ---- CPU baseline: the oracle's C port of the TF1 op chain, bounded sample ---------------
    if rank == 0 and world == 1 and not a.no_cpu_baseline and t_loc > 0:
        from oracle import tf1_path
        pu = plans[0][0]
        S = min(a.cpu_sample_rows, U)
        rp = pu.rowptr[: S + 1].cpu().numpy()
        ne = int(rp[-1])
        ci = pu.colidx[:ne].cpu().numpy()
        idx = np.empty((ne, 2), dtype=np.int32)
        idx[:, 0] = np.repeat(np.arange(S, dtype=np.int32), np.diff(rp))
        idx[:, 1] = ci
        src = emb[0][1].cpu().numpy()
        threads = tf1_path.max_threads()
        scratch = np.empty((max(ne, 1), d), dtype=np.float32)
        tf1_path.message_propagate(idx, src, S, 0.5, threads=threads, scratch=scratch)      # warm-up
        times = []
        while sum(times) < a.cpu_seconds and len(times) < 200:      # a bounded sample: ~10 s of CPU work
            tc = time.perf_counter()
            cpu_out = tf1_path.message_propagate(idx, src, S, 0.5, threads=threads, scratch=scratch)
            times.append(time.perf_counter() - tc)
        sub = ops.SpmmPlan(rp.copy(), ci.copy(), S, I, device=dev, validate=False)
        gpu_out = ops.spmm(sub, emb[0][1], 0.5).cpu().numpy()
        err = float(np.abs(gpu_out - cpu_out).max())
        result["cpu_baseline"] = {
            "value": ne / min(times), "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"user-side SpMM of interval {sh.local_intervals[0]}, rows 0..{S - 1} ({ne} edges), "
                      f"gather->segment_sum->leaky as TF1 runs model.py:86-92 on a CPU; best of {len(times)} passes "
                      f"({sum(times):.1f} s of CPU work, mean {ne / (sum(times) / len(times)) / 1e6:.1f} M edges/s)",
            "seconds": min(times), "gpu_vs_cpu_max_abs_err": err}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
