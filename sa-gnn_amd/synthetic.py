"""Synthetic power-law bipartite interval graphs and a writer for
the reference's on-disk structure `[trnMat, subMat[T], timeMat]` / `sequence`
(reference preprocess_to_trnmat.ipynb cell 7/14-15, preprocess_to_sequence.ipynb cell 6-7).

Generator spec per interval t (seed 1000 + t): user degree ~ Pareto(alpha = 1.8, x_min = 1)
capped at 4096, rescaled so the degrees sum to the edge target; item ~ Zipf(s = 0.8) over a
per-interval random permutation of the items; duplicate (user, item) pairs removed; rows sorted.
Runs on whatever device it is given (the roofline run generates on the GPU). torch is the array
library here, nothing more.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch


def powerlaw_edges(n_users: int, n_items: int, nnz_target: int, seed: int, device="cpu",
                   alpha: float = 1.8, cap: int = 4096, zipf_s: float = 0.8):
    """Returns (user int64 [nnz], item int64 [nnz]) sorted by (user, item), unique, nnz <= target."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    u = torch.rand(n_users, generator=g, device=device, dtype=torch.float64)
    raw = torch.clamp((1.0 - u).clamp_min(1e-12) ** (-1.0 / alpha), max=float(cap))
    # oversample a little: hot items collide inside a user's draw and are removed below
    want = int(nnz_target * 1.12) + 16
    scale = want / float(raw.sum())
    jitter = torch.rand(n_users, generator=g, device=device, dtype=torch.float64)
    deg = torch.floor(raw * scale + jitter).to(torch.int64).clamp_(0, min(cap, n_items))
    users = torch.repeat_interleave(torch.arange(n_users, device=device, dtype=torch.int64), deg)
    m = users.numel()
    # Zipf(s) over ranks 1..I by inverting the continuous CDF  F(r) ~ (r^(1-s) - 1)/(I^(1-s) - 1)
    v = torch.rand(m, generator=g, device=device, dtype=torch.float64)
    one_s = 1.0 - zipf_s
    rank = (v * (float(n_items + 1) ** one_s - 1.0) + 1.0) ** (1.0 / one_s)
    rank = rank.to(torch.int64).clamp_(1, n_items) - 1
    perm = torch.randperm(n_items, generator=g, device=device)
    items = perm[rank]
    key = torch.unique(users * n_items + items)          # sorted, duplicates removed
    if key.numel() > nnz_target:
        keep = torch.randperm(key.numel(), generator=g, device=device)[:nnz_target]
        key = key[torch.sort(keep).values]
    return key // n_items, key % n_items


def csr_pair_from_edges(users: torch.Tensor, items: torch.Tensor, n_users: int, n_items: int):
    """((rowptr_u, colidx_u), (rowptr_i, colidx_i)) int32 tensors on the edges' device.
    Input must be sorted by (user, item) and unique (as powerlaw_edges returns it)."""
    dev = users.device
    zero = torch.zeros(1, dtype=torch.int64, device=dev)
    rp_u = torch.cat([zero, torch.cumsum(torch.bincount(users, minlength=n_users), 0)])
    order = torch.argsort(items * n_users + users)
    rp_i = torch.cat([zero, torch.cumsum(torch.bincount(items, minlength=n_items), 0)])
    return ((rp_u.to(torch.int32), items.to(torch.int32)),
            (rp_i.to(torch.int32), users[order].to(torch.int32)))


def amazon_like_nnz():
    """Per-interval edge counts of Amazon-book (preprocess_to_trnmat.ipynb cell 15 output)."""
    return [72280, 78997, 79692, 78096, 45651]


def make_trn_mat_time(n_users: int, n_items: int, nnz_per_interval, seed0: int = 1000,
                      t0: int = 1_400_000_000, span: int = 30 * 86400):
    """The reference's pickled structure, in memory: [trnMat float64 CSR, [subMat_k intc CSR with
    Unix-timestamp values], timeMat]. trnMat = union pattern (only its .shape is read,
    DataHandler.py:126); timeMat = latest timestamp per pair (stored, never read by the path)."""
    subs = []
    rng = np.random.default_rng(seed0)
    for k, nnz in enumerate(nnz_per_interval):
        u, i = powerlaw_edges(n_users, n_items, int(nnz), seed0 + k)
        ts = (t0 + k * span + rng.integers(0, span, size=u.numel())).astype(np.intc)
        subs.append(sp.csr_matrix((ts, (u.numpy(), i.numpy())), shape=(n_users, n_items), dtype=np.intc))
    union = subs[0].astype(np.float64)
    time_mat = subs[0].copy()
    for s in subs[1:]:
        union = union + s.astype(np.float64)
        time_mat = time_mat.maximum(s)
    union.data[:] = 1.0
    return [sp.csr_matrix(union), subs, sp.csr_matrix(time_mat)]


def make_sequence(trn_mat_time):
    """`sequence`: list of U item lists in time order (preprocess_to_sequence.ipynb cell 6-7)."""
    n_users = trn_mat_time[0].shape[0]
    rows, cols, ts = [], [], []
    for s in trn_mat_time[1]:
        c = s.tocoo()
        rows.append(c.row)
        cols.append(c.col)
        ts.append(c.data)
    rows, cols, ts = np.concatenate(rows), np.concatenate(cols), np.concatenate(ts)
    order = np.lexsort((ts, rows))
    rows, cols = rows[order], cols[order]
    bounds = np.searchsorted(rows, np.arange(n_users + 1))
    return [cols[bounds[u]:bounds[u + 1]].tolist() for u in range(n_users)]


def write_dataset(directory: str, n_users: int, n_items: int, nnz_per_interval, test_size: int = 1000,
                  seed0: int = 1000):
    """Writes a dataset in the reference's on-disk format under `directory` (what
    preprocess_to_trnmat.ipynb / preprocess_to_sequence.ipynb produce): `trn_mat_time`
    ([trnMat, subMat[T], timeMat]), `sequence`, `tst_int` (held-out item per test user, None for
    the others; every second user is a test user) and `test_dict` (1-indexed user -> test_size
    1-indexed candidate items). Returns the in-memory objects too."""
    import os
    import pickle
    os.makedirs(directory, exist_ok=True)
    tmt = make_trn_mat_time(n_users, n_items, nnz_per_interval, seed0)
    seq = make_sequence(tmt)
    rng = np.random.default_rng(seed0 + 7)
    tst_int = [int(rng.integers(0, n_items)) if u % 2 == 0 else None for u in range(n_users)]
    test_dict = {u + 1: [int(v) for v in rng.integers(1, n_items + 1, size=test_size)] for u in range(n_users)}
    for name, obj in (("trn_mat_time", tmt), ("sequence", seq), ("tst_int", tst_int), ("test_dict", test_dict)):
        with open(os.path.join(directory, name), "wb") as fs:
            pickle.dump(obj, fs)
    return tmt, seq, tst_int, test_dict
