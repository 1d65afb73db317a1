"""Interval-parallel execution of the hot path: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The reference is single-process; its loop body model.py:118-129 has no dependency
across intervals, the first cross-interval op is the stack at model.py:131-132. So:

  1. interval k runs on rank k mod world (its CSR pair and embedding slabs live only there);
  2. exchange: every rank sends the row shard r of each of its interval outputs to rank r
     (all-to-all). Rank r ends up with x_r [T, rows_r, d] — all intervals, its rows — received
     straight into place, no staging copy; this moves 1/world of what an all-gather of the
     stacked [T_local, N, d] tensors would (SURVEY.md §8e caveat 1);
  3. fusion (LSTM -> LN -> MHSA -> mean) is row-independent: rank r fuses its rows;
  4. RCCL all-gather reassembles the fused embeddings [N, d] on every rank.

`exchange="allgather"` keeps the plain form of step 2 (all-gather of the stacked interval
outputs, then each rank slices its rows) for comparison.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class IntervalSharding:
    """Static maps: interval -> rank (cyclic) and node row -> rank (contiguous, balanced)."""

    def __init__(self, n_intervals: int, world: int, rank: int):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world {world}")
        self.T, self.world, self.rank = int(n_intervals), int(world), int(rank)
        self.rounds = -(-self.T // self.world)            # all-to-all calls every rank takes part in

    def owner(self, k: int) -> int:
        return k % self.world

    def intervals_of(self, r: int):
        return list(range(r, self.T, self.world))

    @property
    def local_intervals(self):
        return self.intervals_of(self.rank)

    def row_bounds(self, n_rows: int):
        """world+1 offsets; shard r = [b[r], b[r+1]), sizes differ by at most one."""
        q, rem = divmod(int(n_rows), self.world)
        b = [0]
        for r in range(self.world):
            b.append(b[-1] + q + (1 if r < rem else 0))
        return b

    def row_range(self, n_rows: int, r: int | None = None):
        b = self.row_bounds(n_rows)
        r = self.rank if r is None else r
        return b[r], b[r + 1]


def exchange_to_row_shards(local_out: torch.Tensor, sh: IntervalSharding, n_rows: int, group=None,
                           mode: str = "alltoall") -> torch.Tensor:
    """local_out [T_local, N, d]: this rank's interval outputs in local order (interval
    rank + j*world at index j). Returns x [T, rows_local, d] in global interval order."""
    d = local_out.shape[-1]
    if sh.world == 1:
        return local_out                                   # already [T, N, d]; nothing to move
    lo, hi = sh.row_range(n_rows)
    x = torch.empty((sh.T, hi - lo, d), dtype=local_out.dtype, device=local_out.device)
    bounds = sh.row_bounds(n_rows)
    if mode == "alltoall":
        # all_to_all_single per round j: the input is this rank's j-th interval output split by
        # destination row shard; the blocks arrive in sender order s = 0..world-1, i.e. as the
        # global intervals j*world + s, which are adjacent slabs of x — received in place.
        in_splits = [bounds[r + 1] - bounds[r] for r in range(sh.world)]
        for j in range(sh.rounds):
            have = j < local_out.shape[0]
            cnt = min(sh.world, sh.T - j * sh.world)       # senders that own a j-th interval
            out_splits = [(hi - lo) if s < cnt else 0 for s in range(sh.world)]
            recv = x[j * sh.world: j * sh.world + cnt].view(cnt * (hi - lo), d)
            send = local_out[j] if have else local_out.new_empty((0, d))
            dist.all_to_all_single(recv, send, output_split_sizes=out_splits,
                                   input_split_sizes=in_splits if have else [0] * sh.world,
                                   group=group)
        return x
    if mode == "allgather":
        t_loc = sh.rounds
        pad = local_out
        if local_out.shape[0] < t_loc:                    # ranks with one interval fewer pad a slab
            pad = torch.zeros((t_loc, n_rows, d), dtype=local_out.dtype, device=local_out.device)
            pad[:local_out.shape[0]] = local_out
        full = torch.empty((sh.world * t_loc, n_rows, d), dtype=local_out.dtype, device=local_out.device)
        dist.all_gather_into_tensor(full, pad.contiguous(), group=group)
        for k in range(sh.T):
            x[k].copy_(full[(k % sh.world) * t_loc + k // sh.world, lo:hi])
        return x
    raise ValueError(f"unknown exchange mode {mode!r}")


class RowShardExchange:
    """Incremental form of exchange_to_row_shards(mode="alltoall"): round j is posted as soon as
    the rank's j-th interval output exists (async collective on RCCL's own stream), so it moves
    over xGMI while the SpMM stack of interval j+1 runs; finish() makes the current stream wait for
    every posted round and returns x [T, rows_local, d]."""

    def __init__(self, sh: IntervalSharding, n_rows: int, d: int, device, dtype=torch.float32, group=None):
        self.sh, self.n_rows, self.d, self.group = sh, int(n_rows), int(d), group
        self.bounds = sh.row_bounds(n_rows)
        lo, hi = self.bounds[sh.rank], self.bounds[sh.rank + 1]
        self.rows_local = hi - lo
        self.x = torch.empty((sh.T, self.rows_local, d), dtype=dtype, device=device)
        self._empty = torch.empty((0, d), dtype=dtype, device=device)
        self._work = []
        self._posted = 0

    def post(self, out_j: torch.Tensor | None):
        """out_j [N, d]: this rank's next interval output (None when the rank has no interval in
        this round — it still takes part in the collective with empty sends)."""
        sh, j = self.sh, self._posted
        self._posted += 1
        if sh.world == 1:
            self.x[j].copy_(out_j[self.bounds[0]:self.bounds[1]])
            return
        have = out_j is not None
        cnt = min(sh.world, sh.T - j * sh.world)
        out_splits = [self.rows_local if s < cnt else 0 for s in range(sh.world)]
        in_splits = [self.bounds[r + 1] - self.bounds[r] for r in range(sh.world)] if have else [0] * sh.world
        recv = self.x[j * sh.world: j * sh.world + cnt].view(cnt * self.rows_local, self.d)
        w = dist.all_to_all_single(recv, out_j if have else self._empty, output_split_sizes=out_splits,
                                   input_split_sizes=in_splits, group=self.group, async_op=True)
        self._work.append(w)

    def wait_round(self, j: int) -> torch.Tensor:
        """Makes the current stream wait for round j alone and returns its slab
        x[j*world : j*world + cnt] ([cnt, rows_local, d]): the intervals of round j are consecutive
        in time, so the interval LSTM can run their steps while later rounds are still in flight."""
        sh = self.sh
        while self._posted <= j:                   # rounds this rank never had an interval for
            self.post(None)
        if sh.world > 1 and self._work[j] is not None:
            self._work[j].wait()
            self._work[j] = None
        cnt = min(sh.world, sh.T - j * sh.world)
        return self.x[j * sh.world: j * sh.world + cnt]

    def finish(self) -> torch.Tensor:
        while self._posted < self.sh.rounds:       # rounds this rank never had an interval for
            self.post(None)
        for w in self._work:
            if w is not None:
                w.wait()
        self._work.clear()
        self._posted = 0
        return self.x


class RoundFusion:
    """One node type's row-sharded interval fusion (reference model.py:135-155) pipelined with its
    exchange. The intervals of exchange round j are consecutive in time, so `lstm_round(j)` runs
    their LSTM steps as soon as round j has arrived, carrying the state across calls
    (sagnn_lstm_fwd_state_f32: bit-identical to one call over all T); `attention()` then applies
    layer norm + attention + mean. Both take an optional row range, so the tail of the pipeline can
    be cut into row chunks whose all-gathers travel under the next chunk's compute."""

    def __init__(self, ex: "RowShardExchange", p: dict, heads: int, device):
        self.ex, self.p, self.heads, self.device = ex, p, int(heads), device
        sh, rows, d = ex.sh, ex.rows_local, ex.d
        self.h = torch.empty((rows, sh.T, d), dtype=torch.float32, device=device)
        self.c = torch.empty((rows, d), dtype=torch.float32, device=device)

    def lstm_round(self, j: int, lo: int = 0, hi: int | None = None):
        from . import ops
        sh = self.ex.sh
        hi = self.ex.rows_local if hi is None else hi
        xs = self.ex.wait_round(j).to(self.device)[:, lo:hi, :]              # [cnt, rows, d]
        t0, cnt = j * sh.world, xs.shape[0]
        if hi <= lo or cnt == 0:
            return
        h, c = self.h[lo:hi], self.c[lo:hi]
        ops.lstm_fwd(xs.permute(1, 0, 2), self.p["lstm_W"], self.p["lstm_b"], out=h[:, t0:t0 + cnt, :],
                     h0=h[:, t0 - 1, :] if j else None, c0=c if j else None,
                     c_out=c if j + 1 < sh.rounds else None)

    def attention(self, lo: int = 0, hi: int | None = None) -> torch.Tensor:
        from . import ops
        p = self.p
        hi = self.ex.rows_local if hi is None else hi
        return ops.ln_mhsa_mean(self.h[lo:hi], p["ln_gamma"], p["ln_beta"], p["Wq"], p["bq"], p["Wk"], p["bk"],
                                p["Wv"], p["bv"], self.heads)

    def done(self):
        self.ex.finish()                                                    # resets the exchange for the next step


def fuse_as_rounds_arrive(ex: "RowShardExchange", p: dict, heads: int, device):
    """RoundFusion in its simplest order: every round's LSTM steps as the round arrives, then the
    attention; returns the fused rows [rows_local, d]."""
    rf = RoundFusion(ex, p, heads, device)
    for j in range(ex.sh.rounds):
        rf.lstm_round(j)
    out = rf.attention()
    rf.done()
    return out


class ChunkedGather:
    """All-gather of fused rows delivered in row chunks (needs n_rows % world == 0): post(lo, hi,
    rows [hi-lo, d]) starts that chunk's collective at once (async, on RCCL's stream), so it
    travels under whatever is computed next; finish() -> [N, d] in global row order."""

    def __init__(self, sh: IntervalSharding, n_rows: int, group=None):
        if n_rows % sh.world:
            raise ValueError("ChunkedGather needs equal row shards")
        self.sh, self.n_rows, self.group, self.works = sh, int(n_rows), group, []

    def post(self, lo: int, hi: int, f: torch.Tensor):
        buf = torch.empty((self.sh.world * (hi - lo), f.shape[-1]), dtype=f.dtype, device=f.device)
        w = dist.all_gather_into_tensor(buf, f.contiguous(), group=self.group, async_op=True)
        self.works.append((lo, hi, buf, w))

    def finish(self) -> torch.Tensor:
        lo0, hi0, buf0, _ = self.works[0]
        d = buf0.shape[-1]
        out = torch.empty((self.n_rows, d), dtype=buf0.dtype, device=buf0.device)
        view = out.view(self.sh.world, self.n_rows // self.sh.world, d)
        for lo, hi, buf, w in self.works:
            w.wait()
            view[:, lo:hi, :] = buf.view(self.sh.world, hi - lo, d)
        self.works = []
        return out


def gather_fused(final_local: torch.Tensor, sh: IntervalSharding, n_rows: int, group=None,
                 async_op: bool = False):
    """All-gather of the fused embeddings: [rows_local, d] on each rank -> [N, d] everywhere.
    With async_op=True returns (out, finish) — call finish() before reading `out`; the collective
    runs on RCCL's stream meanwhile (e.g. under the other node type's fusion)."""
    d = final_local.shape[-1]
    if sh.world == 1:
        return (final_local, lambda: final_local) if async_op else final_local
    out = torch.empty((n_rows, d), dtype=final_local.dtype, device=final_local.device)
    bounds = sh.row_bounds(n_rows)
    if n_rows % sh.world == 0:
        work = dist.all_gather_into_tensor(out, final_local.contiguous(), group=group, async_op=True)

        def finish():
            work.wait()
            return out
    else:                                                  # pad every shard to the largest one
        rmax = bounds[1] - bounds[0]
        mine = final_local.new_zeros((rmax, d))
        mine[: final_local.shape[0]] = final_local
        full = final_local.new_empty((sh.world * rmax, d))
        work = dist.all_gather_into_tensor(full, mine, group=group, async_op=True)

        def finish():
            work.wait()
            for r in range(sh.world):
                out[bounds[r]:bounds[r + 1]] = full[r * rmax: r * rmax + bounds[r + 1] - bounds[r]]
            return out
    if async_op:
        return out, finish
    return finish()
