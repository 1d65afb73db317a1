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

Fewer intervals than ranks (Amazon T = 5, Gowalla T = 3 on 8 GPUs — SURVEY.md §8e caveat 2): the
ranks are cut into T consecutive GROUPS, sized in proportion to the intervals' edge counts, and the
members of a group split the TARGET ROWS of their interval (`SplitIntervalSharding`,
`SplitIntervalRunner`): every member runs the SpMM of its row slice against the full source table,
the members all-gather the layer output inside the group before the next layer reads it, and the
slice of the running sum goes straight into the same all-to-all (sender order = interval order, so
x [T, rows_r, d] is still received in place). Steps 3 and 4 are unchanged.

Whole-rank groups balance badly when the ranks do not divide by the intervals (Amazon's T = 5 on 8 ranks:
groups [1, 2, 2, 2, 1], busiest rank 72 k edges against 44 k ideal), so the default for T < world is
`FractionalSharding`: the intervals' target rows are laid end to end on one axis weighted by the intervals' edge
counts, and rank r takes the stretch [r, r + 1) / world of it — the tail rows of one interval and the head rows of
the next where the cut falls inside an interval (`FractionalRunner`). Every rank then carries E / world edges (rows
of an interval are taken as equally heavy: user ids and the generator's item permutation are random).
"""
from __future__ import annotations

import torch
import torch.distributed as dist



# bench.py --dist-single sets this: the one-rank shortcuts below are skipped and every collective is issued through a
# one-rank process group — what a one-GPU box can check of the RCCL calls before an 8-GPU node exists
SINGLE_RANK_COLLECTIVES = False


def _solo(sh) -> bool:
    return sh.world == 1 and not SINGLE_RANK_COLLECTIVES


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


class SplitIntervalSharding(IntervalSharding):
    """T < world: rank -> (interval, member index) by consecutive groups. Group sizes follow
    `weights` (edges per interval) by largest remainder, every interval gets at least one rank.
    Inside a group the target rows of a node type are cut into g equal slices of
    q = ceil(n_rows / g) rows (the last one shorter): equal slices keep the intra-group all-gather a
    plain all_gather_into_tensor on a [g*q, d] buffer."""

    def __init__(self, n_intervals: int, world: int, rank: int, weights=None):
        super().__init__(n_intervals, world, rank)
        if not (1 <= self.T < self.world):
            raise ValueError(f"split sharding is for 1 <= T < world, got T={n_intervals}, world={world}")
        w = [1.0] * self.T if weights is None else [float(v) for v in weights]
        if len(w) != self.T or min(w) < 0:
            raise ValueError("weights: one non-negative number per interval")
        tot = sum(w) or float(self.T)
        spare = self.world - self.T                         # ranks beyond one per interval
        ideal = [spare * (v / tot if sum(w) else 1.0 / self.T) for v in w]
        extra = [int(x) for x in ideal]
        order = sorted(range(self.T), key=lambda k: (-(ideal[k] - extra[k]), k))
        for k in order[: spare - sum(extra)]:
            extra[k] += 1
        self.group_size = [1 + e for e in extra]
        self.group_first = [sum(self.group_size[:k]) for k in range(self.T)]
        self.rounds = 1
        k = max(i for i in range(self.T) if self.group_first[i] <= self.rank)
        self.interval, self.member = k, self.rank - self.group_first[k]

    def owner(self, k: int) -> int:
        return self.group_first[k]                           # first member

    def members(self, k: int):
        return list(range(self.group_first[k], self.group_first[k] + self.group_size[k]))

    def intervals_of(self, r: int):
        return [max(i for i in range(self.T) if self.group_first[i] <= r)]

    def slice_rows(self, n_rows: int, k: int | None = None):
        """q = rows per member slice of interval k (padded length of the group buffer is g*q)."""
        g = self.group_size[self.interval if k is None else k]
        return -(-int(n_rows) // g)

    def slice_range(self, n_rows: int, r: int | None = None):
        """[lo, hi) target rows rank r computes (may be empty for the last members of tiny matrices)."""
        r = self.rank if r is None else r
        k = self.intervals_of(r)[0]
        q = self.slice_rows(n_rows, k)
        m = r - self.group_first[k]
        return min(m * q, int(n_rows)), min((m + 1) * q, int(n_rows))

    def exchange_splits(self, n_rows: int):
        """(input_split_sizes, output_split_sizes) in rows of the ONE all-to-all: what this rank's
        slice sends to each row shard, and what it receives from each sender's slice."""
        b = self.row_bounds(n_rows)
        ov = lambda a0, a1, b0, b1: max(0, min(a1, b1) - max(a0, b0))     # noqa: E731
        lo, hi = self.slice_range(n_rows)
        ins = [ov(lo, hi, b[r], b[r + 1]) for r in range(self.world)]
        mine = (b[self.rank], b[self.rank + 1])
        outs = [ov(*self.slice_range(n_rows, s), *mine) for s in range(self.world)]
        return ins, outs


class FractionalSharding(IntervalSharding):
    """T < world with edge-balanced cuts: the T intervals lie end to end on [0, 1) in proportion to `weights`
    (edges per interval); rank r owns [r / world, (r + 1) / world). Its SEGMENTS are the intervals that stretch
    meets, each with the fraction range (f0, f1) of the interval's target rows it computes — the same fractions for
    the user side and the item side. The ranks that meet interval k are consecutive (`members(k)`), a rank may sit in
    two neighbouring groups. Exact rational arithmetic on integer weights: every rank derives the same cuts."""

    def __init__(self, n_intervals: int, world: int, rank: int, weights=None):
        super().__init__(n_intervals, world, rank)
        if not (1 <= self.T < self.world):
            raise ValueError(f"fractional sharding is for 1 <= T < world, got T={n_intervals}, world={world}")
        w = [1] * self.T if weights is None else [int(v) for v in weights]
        if len(w) != self.T or min(w) < 0:
            raise ValueError("weights: one non-negative integer per interval")
        if sum(w) == 0:
            w = [1] * self.T
        self.w, self.W = w, sum(w)
        self.cum = [0]
        for v in w:
            self.cum.append(self.cum[-1] + v)
        self.rounds = 1
        self._segs = [self._segments_of(r) for r in range(self.world)]

    def _segments_of(self, r: int):
        # everything in units of 1 / (W * world): rank r = [r W, (r + 1) W), interval k = [cum_k world, cum_{k+1} world)
        lo, hi = r * self.W, (r + 1) * self.W
        segs = []
        for k in range(self.T):
            a, b = self.cum[k] * self.world, self.cum[k + 1] * self.world
            x0, x1 = max(lo, a), min(hi, b)
            if x1 > x0 and b > a:
                segs.append((k, (x0 - a, b - a), (x1 - a, b - a)))       # fractions as (numerator, denominator)
        return segs

    def segments(self, r: int | None = None):
        """[(interval, member index in its group)] of rank r, ascending in the interval."""
        r = self.rank if r is None else r
        return [(k, r - self.members(k)[0]) for k, _, _ in self._segs[r]]

    def members(self, k: int):
        return [r for r in range(self.world) if any(s[0] == k for s in self._segs[r])]

    def intervals_of(self, r: int):
        return [s[0] for s in self._segs[r]]

    def owner(self, k: int) -> int:
        return self.members(k)[0]

    def cuts(self, n_rows: int, k: int):
        """g + 1 row offsets of interval k's member slices ([0 .. n_rows], monotone)."""
        mem = self.members(k)
        b = [0]
        for r in mem:
            (_, _, (num, den)) = next(s for s in self._segs[r] if s[0] == k)
            b.append((num * int(n_rows)) // den)
        b[-1] = int(n_rows)
        return b

    def slice_range(self, n_rows: int, k: int, r: int | None = None):
        r = self.rank if r is None else r
        mem = self.members(k)
        c = self.cuts(n_rows, k)
        m = mem.index(r)
        return c[m], c[m + 1]

    def exchange_splits(self, n_rows: int):
        """(ins, outs) of the ONE all-to-all in rows: what this rank's segments (in interval order) send to each row
        shard, what it receives from each sender (whose block is ordered (interval, row), which is the order of
        x [T, rows_local, d] because ranks are monotone in the interval)."""
        b = self.row_bounds(n_rows)
        ov = lambda a0, a1, b0, b1: max(0, min(a1, b1) - max(a0, b0))     # noqa: E731
        ins = [sum(ov(*self.slice_range(n_rows, k), b[r], b[r + 1]) for k in self.intervals_of(self.rank)) for r in range(self.world)]
        mine = (b[self.rank], b[self.rank + 1])
        outs = [sum(ov(*self.slice_range(n_rows, k, s), *mine) for k in self.intervals_of(s)) for s in range(self.world)]
        return ins, outs

    def send_order(self, n_rows: int):
        """Row pieces of this rank's concatenated segment rows in SEND order (destination shard, then interval):
        [(offset, length)] into the concatenation; all_to_all_single wants the input grouped by destination."""
        b = self.row_bounds(n_rows)
        base, segs = 0, []
        for k in self.intervals_of(self.rank):
            lo, hi = self.slice_range(n_rows, k)
            segs.append((base, lo, hi))
            base += hi - lo
        pieces = []
        for r in range(self.world):
            for off, lo, hi in segs:
                a0, a1 = max(lo, b[r]), min(hi, b[r + 1])
                if a1 > a0:
                    pieces.append((off + a0 - lo, a1 - a0))
        return pieces


def make_sharding(n_intervals: int, world: int, rank: int, weights=None, split: str = "fractional") -> IntervalSharding:
    """Cyclic interval sharding when every rank gets an interval; for T < world the edge-balanced fractional cuts
    (split="fractional", the default) or whole-rank groups (split="groups")."""
    if 1 <= n_intervals < world:
        if split == "groups":
            return SplitIntervalSharding(n_intervals, world, rank, weights)
        return FractionalSharding(n_intervals, world, rank, None if weights is None else [int(v) for v in weights])
    return IntervalSharding(n_intervals, world, rank)


def _all_gather_rows(table: torch.Tensor, cuts, m: int, group, comm_device=None):
    """table [N, d] with this member's rows [cuts[m], cuts[m+1]) up to date -> every member's rows. Slices differ in
    length: RCCL takes the uneven list form directly (grouped broadcasts); gloo needs equal pieces, so the gloo
    path pads to the longest slice."""
    g = len(cuts) - 1
    if g == 1:
        return
    staged = comm_device is not None and torch.device(comm_device) != table.device
    if dist.get_backend(group) != "gloo" and not staged:
        outs = [table[cuts[j]:cuts[j + 1]] for j in range(g)]
        dist.all_gather(outs, table[cuts[m]:cuts[m + 1]].clone(), group=group)
        return
    q = max(cuts[j + 1] - cuts[j] for j in range(g))
    dev = table.device if not staged else torch.device(comm_device)
    mine = torch.zeros((q, table.shape[1]), dtype=table.dtype, device=dev)
    mine[: cuts[m + 1] - cuts[m]] = table[cuts[m]:cuts[m + 1]].to(dev)
    full = torch.empty((g * q, table.shape[1]), dtype=table.dtype, device=dev)
    dist.all_gather_into_tensor(full, mine, group=group)
    for j in range(g):
        if j != m:
            table[cuts[j]:cuts[j + 1]] = full[j * q: j * q + cuts[j + 1] - cuts[j]].to(table.device)


class FractionalRunner:
    """The L-layer stacks (reference model.py:118-129) of the intervals a rank meets under FractionalSharding, each on
    the rank's slice of the target rows. plans[k] = (plan_u, plan_i) of THIS rank's row slices of interval k
    (csr_row_slice) against the full source tables; groups[k] = the process group of interval k's members (None on
    ranks outside it). After run(): out_u / out_i = the rank's slices of sum_l e^l, concatenated in interval order
    (what RowShardExchange.post takes). Layer outputs are all-gathered inside each interval's group, ascending in
    the interval — a rank in two groups takes part in both, in the same order as everybody else."""

    def __init__(self, sh: FractionalSharding, n_users: int, n_items: int, d: int, device, groups, dtype=torch.float32,
                 comm_device=None):
        self.sh, self.U, self.I, self.d, self.groups = sh, int(n_users), int(n_items), int(d), groups
        self.comm_device = comm_device
        self.ks = sh.intervals_of(sh.rank)
        self.tab_u = {k: torch.zeros((2, self.U, d), dtype=dtype, device=device) for k in self.ks}
        self.tab_i = {k: torch.zeros((2, self.I, d), dtype=dtype, device=device) for k in self.ks}
        self.ru = {k: sh.slice_range(self.U, k) for k in self.ks}
        self.ri = {k: sh.slice_range(self.I, k) for k in self.ks}
        self.out_u = torch.empty((sum(hi - lo for lo, hi in self.ru.values()), d), dtype=dtype, device=device)
        self.out_i = torch.empty((sum(hi - lo for lo, hi in self.ri.values()), d), dtype=dtype, device=device)

    def _views(self, buf_u, buf_i):
        ou, oi, v = 0, 0, {}
        for k in self.ks:
            (lu, hu), (li, hi) = self.ru[k], self.ri[k]
            v[k] = (buf_u[ou: ou + hu - lu], buf_i[oi: oi + hi - li])
            ou, oi = ou + hu - lu, oi + hi - li
        return v

    def run(self, spmm, plans: dict, emb: dict, n_layers: int, leaky: float, masks: dict | None = None):
        """emb[k] = (u0 [U, d], i0 [I, d]) of every interval this rank meets. masks[k] = (mask_u [L, rows_u, d/4],
        mask_i [L, rows_i, d/4]) uint8: record the activation slopes of this rank's rows (training; spmm must then take
        the extended epilogue, ops.spmm_ex)."""
        sh = self.sh
        acc = self._views(self.out_u, self.out_i)     # the rank's slices of the running sums, views into the send buffers
        cur = {k: emb[k] for k in self.ks}
        for l in range(n_layers):
            last = l + 1 == n_layers
            for k in self.ks:
                (lu, hu), (li, hi) = self.ru[k], self.ri[k]
                cu, ci = cur[k]
                nu, ni = self.tab_u[k][l & 1], self.tab_i[k][l & 1]
                for side, (plan, src, c, lo, hi_, nxt, a) in enumerate(((plans[k][0], ci, cu, lu, hu, nu, acc[k][0]),
                                                                        (plans[k][1], cu, ci, li, hi, ni, acc[k][1]))):
                    if hi_ > lo:
                        kw = {} if masks is None else {"mask_out": masks[k][side][l]}
                        spmm(plan, src[: plan.n_src], leaky, residual=c[lo:hi_], out=None if last else nxt[lo:hi_],
                             acc_in=c[lo:hi_] if l == 0 else a, acc_out=a, want_out=not last, **kw)
            if last:
                break
            for k in self.ks:                    # ascending k on every rank: no cycle between overlapping groups
                m = sh.members(k).index(sh.rank)
                _all_gather_rows(self.tab_u[k][l & 1], sh.cuts(self.U, k), m, self.groups[k], self.comm_device)
                _all_gather_rows(self.tab_i[k][l & 1], sh.cuts(self.I, k), m, self.groups[k], self.comm_device)
                cur[k] = (self.tab_u[k][l & 1], self.tab_i[k][l & 1])
        return self.out_u, self.out_i

    def run_backward(self, spmm, mask_scale, plans: dict, G_u: torch.Tensor, G_i: torch.Tensor, masks: dict, n_layers: int,
                     leaky: float):
        """Backward of run(): G_u / G_i = dL/d(out_u), dL/d(out_i) (the rank's slices, interval order) -> {k: (dL/d u0
        [U, d], dL/d i0 [I, d])} FULL tables, the same on every member of interval k's group. The recurrence of
        sagnn_gnn_interval_bwd_f32 on row slices:  g^l[slice] = G + g^{l+1}[slice] + A[slice, :] (g_partner^{l+1} * m^{l+1}),
        where the masked partner gradient is a FULL table: every member makes its slice of it (the epilogue's out2) and
        the group all-gathers it — exactly the forward's pattern, on the forward's own row-slice plans (matrices without
        duplicated stored entries: the user-side pattern is the adjoint of the item-side one)."""
        sh, d = self.sh, self.d
        Gv = self._views(G_u, G_i)
        dev, dt = G_u.device, G_u.dtype
        gfull = {k: tuple(torch.empty((2,) + tuple(g.shape), dtype=dt, device=dev) for g in Gv[k]) for k in self.ks}
        out = {}
        for k in self.ks:                        # seed: (G * m^L) of my rows into the full tables, then the group's
            (lu, hu), (li, hi) = self.ru[k], self.ri[k]
            if hu > lu:
                mask_scale(Gv[k][0], masks[k][0][n_layers - 1], leaky, self.tab_u[k][0][lu:hu])
            if hi > li:
                mask_scale(Gv[k][1], masks[k][1][n_layers - 1], leaky, self.tab_i[k][0][li:hi])
        cur = 0
        gnext = {k: Gv[k] for k in self.ks}      # g^{l+1} slices
        for l in range(n_layers - 1, -1, -1):
            for k in self.ks:
                m = sh.members(k).index(sh.rank)
                _all_gather_rows(self.tab_u[k][cur], sh.cuts(self.U, k), m, self.groups[k], self.comm_device)
                _all_gather_rows(self.tab_i[k][cur], sh.cuts(self.I, k), m, self.groups[k], self.comm_device)
            final = l == 0
            for k in self.ks:
                (lu, hu), (li, hi) = self.ru[k], self.ri[k]
                if final:
                    out[k] = (torch.zeros((self.U, d), dtype=dt, device=dev), torch.zeros((self.I, d), dtype=dt, device=dev))
                for side, (plan, src, lo, hi_) in enumerate(((plans[k][0], self.tab_i[k][cur], lu, hu),
                                                             (plans[k][1], self.tab_u[k][cur], li, hi))):
                    if hi_ <= lo:
                        continue
                    dst = out[k][side][lo:hi_] if final else gfull[k][side][l & 1]
                    kw = {}
                    if not final:
                        tab = (self.tab_u if side == 0 else self.tab_i)[k][cur ^ 1]
                        kw = {"mask_in": masks[k][side][l - 1], "out2": tab[lo:hi_], "slope2": leaky}
                    spmm(plan, src[: plan.n_src], 1.0, residual=gnext[k][side], acc_in=Gv[k][side], acc_out=dst, want_out=False, **kw)
                if not final:
                    gnext[k] = (gfull[k][0][l & 1], gfull[k][1][l & 1])
            cur ^= 1
        for k in self.ks:                        # every member ends with the whole gradient of the replicated tables
            m = sh.members(k).index(sh.rank)
            _all_gather_rows(out[k][0], sh.cuts(self.U, k), m, self.groups[k], self.comm_device)
            _all_gather_rows(out[k][1], sh.cuts(self.I, k), m, self.groups[k], self.comm_device)
        return out


class FractionalStackFn(torch.autograd.Function):
    """FractionalRunner.run with its adjoint: (u0, i0 of every interval the rank meets; replicated inside an interval's
    group) -> (out_u, out_i) = the rank's row slices of sum_l e^l. Backward returns the WHOLE gradient of every table
    (all-gathered inside the group), so that every member applies the same optimiser step to its replica."""

    @staticmethod
    def forward(ctx, runner, spmm, mask_scale, plans, n_layers, leaky, *embs):
        ks = runner.ks
        emb = {k: (embs[2 * j].detach(), embs[2 * j + 1].detach()) for j, k in enumerate(ks)}
        d, dev = runner.d, embs[0].device
        masks = {k: (torch.empty((n_layers, runner.ru[k][1] - runner.ru[k][0], d // 4), dtype=torch.uint8, device=dev),
                     torch.empty((n_layers, runner.ri[k][1] - runner.ri[k][0], d // 4), dtype=torch.uint8, device=dev)) for k in ks}
        ou, oi = runner.run(spmm, plans, emb, n_layers, leaky, masks=masks)
        ctx.runner, ctx.spmm, ctx.mask_scale, ctx.plans, ctx.cfg, ctx.masks = runner, spmm, mask_scale, plans, (n_layers, leaky), masks
        return ou.clone(), oi.clone()

    @staticmethod
    def backward(ctx, g_u, g_i):
        n_layers, leaky = ctx.cfg
        grads = ctx.runner.run_backward(ctx.spmm, ctx.mask_scale, ctx.plans, g_u.contiguous(), g_i.contiguous(), ctx.masks, n_layers,
                                        leaky)
        flat = []
        for k in ctx.runner.ks:
            flat += [grads[k][0], grads[k][1]]
        return (None, None, None, None, None, None) + tuple(flat)


def csr_row_slice(rowptr, colidx, lo: int, hi: int):
    """CSR arrays of rows [lo, hi) with all columns (numpy or torch int32 in, same kind out)."""
    base = int(rowptr[lo])
    return rowptr[lo:hi + 1] - base, colidx[base:int(rowptr[hi])]


class SplitIntervalRunner:
    """One interval's L-layer stack (reference model.py:118-129) computed by the g members of a
    group, each on its slice of the target rows.

    spmm(plan, x, leaky, residual=, out=, acc_in=, acc_out=, want_out=) is ops.spmm (tests pass a
    CPU stand-in: this class is sharding logic, not arithmetic). plan_u / plan_i are plans of THIS
    member's row slices (csr_row_slice) against the full source tables. After run(), acc_u / acc_i
    hold the member's slice of sum_l e^l; layer outputs are all-gathered inside the group (L-1 times
    per node type) because the next layer gathers from every row of the partner table."""

    def __init__(self, sh: SplitIntervalSharding, n_users: int, n_items: int, d: int, device, group=None,
                 dtype=torch.float32, comm_device=None):
        self.sh, self.U, self.I, self.d, self.group = sh, int(n_users), int(n_items), int(d), group
        # comm_device != device: collectives run on host copies (the gloo rehearsal on one GPU)
        self.comm_device = torch.device(device) if comm_device is None else torch.device(comm_device)
        self.staged = self.comm_device != torch.device(device)
        self.g = sh.group_size[sh.interval]
        self.qu, self.qi = sh.slice_rows(n_users), sh.slice_rows(n_items)
        self.ru, self.ri = sh.slice_range(n_users), sh.slice_range(n_items)
        # full-table ping-pong buffers padded to g*q rows; row m*q + j = row j of member m's slice
        self.buf_u = torch.zeros((2, self.g * self.qu, d), dtype=dtype, device=device)
        self.buf_i = torch.zeros((2, self.g * self.qi, d), dtype=dtype, device=device)
        self.acc_u = torch.empty((self.ru[1] - self.ru[0], d), dtype=dtype, device=device)
        self.acc_i = torch.empty((self.ri[1] - self.ri[0], d), dtype=dtype, device=device)

    def run(self, spmm, plan_u, plan_i, u0: torch.Tensor, i0: torch.Tensor, n_layers: int, leaky: float):
        sh, m = self.sh, self.sh.member
        cur_u, cur_i = u0, i0                                  # full [U, d] / [I, d]
        (lu, hu), (li, hi) = self.ru, self.ri
        for l in range(n_layers):
            last = l + 1 == n_layers
            nu, ni = self.buf_u[l & 1], self.buf_i[l & 1]
            mine_u = nu[m * self.qu: m * self.qu + (hu - lu)]
            mine_i = ni[m * self.qi: m * self.qi + (hi - li)]
            for plan, src, cur, lo, hi_, mine, acc in ((plan_u, cur_i, cur_u, lu, hu, mine_u, self.acc_u),
                                                       (plan_i, cur_u, cur_i, li, hi, mine_i, self.acc_i)):
                if hi_ > lo:
                    spmm(plan, src[: plan.n_src], leaky, residual=cur[lo:hi_], out=None if last else mine,
                         acc_in=cur[lo:hi_] if l == 0 else acc, acc_out=acc, want_out=not last)
            if last:
                break
            if self.g > 1:                                     # e^{l+1} of every member, in slice order
                for buf, q in ((nu, self.qu), (ni, self.qi)):
                    if self.staged:
                        full = torch.empty(buf.shape, dtype=buf.dtype, device=self.comm_device)
                        dist.all_gather_into_tensor(full, buf[m * q:(m + 1) * q].to(self.comm_device), group=self.group)
                        buf.copy_(full)
                    else:
                        dist.all_gather_into_tensor(buf, buf[m * q:(m + 1) * q].clone(), group=self.group)
            cur_u, cur_i = nu[: self.U], ni[: self.I]           # g*q >= n_rows and slices are contiguous
        return self.acc_u, self.acc_i


def exchange_to_row_shards(local_out: torch.Tensor, sh: IntervalSharding, n_rows: int, group=None,
                           mode: str = "alltoall") -> torch.Tensor:
    """local_out [T_local, N, d]: this rank's interval outputs in local order (interval
    rank + j*world at index j). Returns x [T, rows_local, d] in global interval order."""
    d = local_out.shape[-1]
    if _solo(sh):
        return local_out                                   # already [T, N, d]; nothing to move
    lo, hi = sh.row_range(n_rows)
    x = torch.empty((sh.T, hi - lo, d), dtype=local_out.dtype, device=local_out.device)
    bounds = sh.row_bounds(n_rows)
    if isinstance(sh, (SplitIntervalSharding, FractionalSharding)):
        # local_out: this rank's row slice(s) [rows, d] of its interval output(s). One all-to-all;
        # senders arrive in rank order = (interval, slice) order = the layout of x.
        if mode != "alltoall":
            raise ValueError("split sharding (T < world) exchanges by all-to-all only")
        ins, outs = sh.exchange_splits(n_rows)
        dist.all_to_all_single(x.view(sh.T * (hi - lo), d), _send_rows(local_out.reshape(-1, d), sh, n_rows),
                               output_split_sizes=outs, input_split_sizes=ins, group=group)
        return x
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


def _send_rows(rows: torch.Tensor, sh, n_rows: int) -> torch.Tensor:
    """The rank's slice rows in SEND order (grouped by destination shard). One segment: already so."""
    if not isinstance(sh, FractionalSharding) or len(sh.intervals_of(sh.rank)) <= 1:
        return rows
    pieces = sh.send_order(n_rows)
    return torch.cat([rows[o:o + n] for o, n in pieces]) if pieces else rows[:0]


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
        self._split = sh.exchange_splits(n_rows) if isinstance(sh, (SplitIntervalSharding, FractionalSharding)) else None

    def post(self, out_j: torch.Tensor | None):
        """out_j [N, d]: this rank's next interval output (None when the rank has no interval in
        this round — it still takes part in the collective with empty sends)."""
        sh, j = self.sh, self._posted
        self._posted += 1
        if _solo(sh):
            self.x[j].copy_(out_j[self.bounds[0]:self.bounds[1]])
            return
        if self._split is not None:           # T < world: out_j is this member's row slice of its interval
            ins, outs = self._split
            w = dist.all_to_all_single(self.x.view(sh.T * self.rows_local, self.d),
                                       _send_rows(out_j.reshape(-1, self.d), sh, self.n_rows),
                                       output_split_sizes=outs, input_split_sizes=ins, group=self.group, async_op=True)
            self._work.append(w)
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
        if self._split is not None:
            return self.x
        cnt = min(sh.world, sh.T - j * sh.world)
        return self.x[j * sh.world: j * sh.world + cnt]

    def slab(self, j: int) -> torch.Tensor:
        """Round j's slab of x as it stands, without posting or waiting (timing passes on data already received)."""
        sh = self.sh
        if self._split is not None:
            return self.x
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

    def lstm_round(self, j: int, lo: int = 0, hi: int | None = None, wait: bool = True):
        from . import ops
        sh = self.ex.sh
        hi = self.ex.rows_local if hi is None else hi
        xs = (self.ex.wait_round(j) if wait else self.ex.slab(j)).to(self.device)[:, lo:hi, :]   # [cnt, rows, d]
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
    if _solo(sh):
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


# ----------------------------------------------------------------------------------------------
# Backward of the exchange: what tf.gradients would add if the reference were sharded this way.
# all-gather  <->  reduce-scatter;  all-to-all into row shards  <->  the reverse all-to-all.
# ----------------------------------------------------------------------------------------------


def _reduce_scatter_rows(g: torch.Tensor, sh: IntervalSharding, n_rows: int, group=None) -> torch.Tensor:
    """sum over ranks of g [N, d], each rank keeping its row shard [rows_local, d]."""
    lo, hi = sh.row_range(n_rows)
    if _solo(sh):
        return g[lo:hi]
    g = g.contiguous()
    if dist.get_backend(group) == "gloo":          # gloo has no reduce-scatter: all-reduce, keep the shard
        full = g.clone()
        dist.all_reduce(full, group=group)
        return full[lo:hi].contiguous()
    d = g.shape[-1]
    bounds = sh.row_bounds(n_rows)
    if n_rows % sh.world == 0:
        out = g.new_empty((hi - lo, d))
        dist.reduce_scatter_tensor(out, g, group=group)
        return out
    rmax = bounds[1] - bounds[0]                    # pad every shard to the largest one
    padded = g.new_zeros((sh.world * rmax, d))
    for r in range(sh.world):
        padded[r * rmax: r * rmax + bounds[r + 1] - bounds[r]] = g[bounds[r]:bounds[r + 1]]
    out = g.new_empty((rmax, d))
    dist.reduce_scatter_tensor(out, padded, group=group)
    return out[: hi - lo].contiguous()


class GatherFusedFn(torch.autograd.Function):
    """gather_fused with its adjoint: forward all-gathers the fused rows [rows_local, d] -> [N, d]; backward
    reduce-scatters dL/dF (every rank may hold a different contribution: its own batch of users) back to the
    row shards — the 'matching reduce-scatter' of SURVEY.md §8(e)."""

    @staticmethod
    def forward(ctx, f_local, sh, n_rows, group):
        ctx.sh, ctx.n_rows, ctx.group = sh, n_rows, group
        return gather_fused(f_local.detach(), sh, n_rows, group)

    @staticmethod
    def backward(ctx, g):
        return _reduce_scatter_rows(g, ctx.sh, ctx.n_rows, ctx.group), None, None, None


class ExchangeRowsFn(torch.autograd.Function):
    """exchange_to_row_shards(mode="alltoall") with its adjoint: forward [T_local, N, d] (T >= world) or the rank's row
    slices [rows, d] (T < world) -> x [T, rows_local, d]; backward sends every gradient row back to where its value came
    from (the same all-to-all with send and receive swapped)."""

    @staticmethod
    def forward(ctx, local_out, sh, n_rows, group):
        ctx.sh, ctx.n_rows, ctx.group, ctx.t_loc = sh, n_rows, group, local_out.shape[0]
        return exchange_to_row_shards(local_out.detach().contiguous(), sh, n_rows, group, mode="alltoall")

    @staticmethod
    def backward(ctx, g):
        sh, n_rows = ctx.sh, ctx.n_rows
        d = g.shape[-1]
        if _solo(sh):
            return g, None, None, None
        g = g.contiguous()
        bounds = sh.row_bounds(n_rows)
        rows_local = bounds[sh.rank + 1] - bounds[sh.rank]
        if isinstance(sh, (SplitIntervalSharding, FractionalSharding)):
            ins, outs = sh.exchange_splits(n_rows)               # forward: sent `ins`, received `outs`
            back = g.new_empty((sum(ins), d))                    # arrives in SEND order (grouped by the shard it went to)
            dist.all_to_all_single(back, g.view(sh.T * rows_local, d), output_split_sizes=ins, input_split_sizes=outs,
                                   group=ctx.group)
            if isinstance(sh, FractionalSharding) and len(sh.intervals_of(sh.rank)) > 1:
                d_rows = torch.empty_like(back)
                pos = 0
                for off, n in sh.send_order(n_rows):             # undo the send permutation
                    d_rows[off:off + n] = back[pos:pos + n]
                    pos += n
                back = d_rows
            return back, None, None, None
        d_local = g.new_empty((ctx.t_loc, n_rows, d))
        shard_sizes = [bounds[r + 1] - bounds[r] for r in range(sh.world)]
        for j in range(sh.rounds):
            have = j < ctx.t_loc
            cnt = min(sh.world, sh.T - j * sh.world)              # owners that hold a j-th interval
            send = g[j * sh.world: j * sh.world + cnt].reshape(cnt * rows_local, d)
            recv = d_local[j] if have else g.new_empty((0, d))
            dist.all_to_all_single(recv, send, output_split_sizes=shard_sizes if have else [0] * sh.world,
                                   input_split_sizes=[rows_local if s < cnt else 0 for s in range(sh.world)],
                                   group=ctx.group)
        return d_local, None, None, None


def exchange_rows(local_out: torch.Tensor, sh: IntervalSharding, n_rows: int, group=None) -> torch.Tensor:
    return ExchangeRowsFn.apply(local_out, sh, n_rows, group)


def gather_rows(f_local: torch.Tensor, sh: IntervalSharding, n_rows: int, group=None) -> torch.Tensor:
    return GatherFusedFn.apply(f_local, sh, n_rows, group)


def allreduce_grads(tensors, group=None):
    """Sums the gradients of replicated parameters (the fusion weights: every rank saw only its rows)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in tensors:
        if t.grad is not None:
            dist.all_reduce(t.grad, group=group)
