#!/usr/bin/env python3
"""Generates the issue schedules of the hand-scheduled f16 LSTM kernels: the order in which the gate-math
operations of one 16-row batch tile (numbered as in `gate_op` of the kernel headers) are issued between the
MFMAs of the next tile, and where each MFMA gap's share begins.

    python tools/gen_lstm_schedule.py 4 > sa-gnn_amd/csrc/lstm_f16_schedule.inc     # 4 hidden units per lane
    python tools/gen_lstm_schedule.py 2                                            # 2 hidden units per lane: the eight-waves-per-
                                                                                    # workgroup form measured as no gain

A list scheduler over the operations' dependency graph: every MFMA gap gets at most `max_trans`
transcendentals (v_exp_f32 / v_rcp_f32: 8 issue cycles, the rest 4 — MI355X_MICROARCH.md, per-instruction
constants) and plain operations up to an even share of the total issue cost; an operation is ready
when everything it reads was issued in an EARLIER gap (>= 16 cycles before: no result latency is exposed).
Critical-path-first. One table per (gaps per tile, with / without the operations that split the tile's
share of the next step's x): U = 4: 48 / 24 / 12 gaps (d = 64 recurrent step, its first step = d = 32
recurrent, d = 32 first step) and two x passes; U = 2: 24 / 12 / 6 gaps and one x pass.

Operation numbers for U hidden units per lane (U = 4 is the list in lstm_f16_kernel.h):
  0 dropout-mask load | 1.. join of the accumulators (4U) | exp2 (4U) | 1 + e (4U) | rcp (4U) | tanh(j) fix (U) |
  i tanh(j) (U) | c' (U) | c' 2 log2 e (U) | exp2 (U) | 1 + e (U) | rcp (U) | tanh(c') fix (U) | h (U) | dropout (U) |
  heads of h (U/2) | residuals (U) | tails (U) | LDS write | h store + c | stores of the training forward"""
import sys

K_X = 16


class Ops:
    """Operation numbers of `gate_op` in lstm_f16_kernel.h (4 hidden units per lane, element-wise steps on pairs)."""
    def __init__(self):
        self.join, self.exp, self.add1, self.rcp = 1, 9, 25, 33          # 8 pk, 16 trans, 8 pk, 16 trans
        self.tj, self.pr, self.cn, self.um = 49, 51, 53, 55                 # 2 pk each
        self.ue, self.ua, self.ur, self.ut = 57, 61, 63, 67                 # 4 trans, 2 pk, 4 trans, 2 pk
        self.hn, self.hv, self.head, self.res, self.tail = 69, 71, 73, 75, 79
        self.write, self.store, self.save, self.n_gate = 83, 84, 85, 86
        self.trans = set(range(9, 25)) | set(range(33, 49)) | set(range(57, 61)) | set(range(63, 67))
        self.packed = (set(range(1, 9)) | set(range(25, 33)) | set(range(49, 57)) | set(range(61, 63)) | set(range(67, 73)))


def build(xpasses):
    o = Ops()
    n = o.n_gate + xpasses * K_X
    deps = {k: set() for k in range(n)}
    for g in range(4):
        for r in range(4):
            e = o.exp + 4 * g + r
            deps[e].add(o.join + 2 * g + r // 2)
            deps[o.add1 + 2 * g + r // 2].add(e)
            deps[o.rcp + 4 * g + r].add(o.add1 + 2 * g + r // 2)
    for p in range(2):
        rc = lambda g: {o.rcp + 4 * g + 2 * p, o.rcp + 4 * g + 2 * p + 1}     # noqa: E731
        deps[o.tj + p] |= rc(1)
        deps[o.pr + p] |= rc(0) | {o.tj + p}
        deps[o.cn + p] |= rc(2) | {o.pr + p}
        deps[o.um + p].add(o.cn + p)
        for r in (2 * p, 2 * p + 1):
            deps[o.ue + r].add(o.um + p)
            deps[o.ua + p].add(o.ue + r)
            deps[o.ur + r].add(o.ua + p)
            deps[o.ut + p].add(o.ur + r)
        deps[o.hn + p] |= {o.ut + p} | rc(3)
        deps[o.hv + p] |= {o.hn + p, 0}
        deps[o.head + p].add(o.hn + p)
        for r in (2 * p, 2 * p + 1):
            deps[o.res + r] |= {o.head + p, o.hn + p}
            deps[o.tail + r].add(o.res + r)
        deps[o.tail + 2 * p + 1].add(o.tail + 2 * p)          # v_fma_mixhi_f16 into the register v_fma_mixlo_f16 wrote
    deps[o.write] |= {o.head, o.head + 1} | set(range(o.tail, o.tail + 4))
    deps[o.store] |= {o.hv, o.hv + 1, o.cn, o.cn + 1}
    deps[o.save] |= set(range(o.rcp, o.rcp + 16)) | {o.tj, o.tj + 1, o.cn, o.cn + 1}
    for p in range(xpasses):
        b = o.n_gate + p * K_X
        for i in range(4):
            deps[b + 2 + i].add(b + i // 2)
        deps[b + 6].add(b + 2)
        deps[b + 7] |= {b + 6, b + 3}
        deps[b + 8].add(b + 4)
        deps[b + 9] |= {b + 8, b + 5}
        deps[b + 11].add(b + 10)                          # the segment's max |v| ...
        deps[b + 12].add(b + 11)                          # ... into the running max,
        deps[b + 13].add(b + 11)                          # ... its bits - 1
        deps[b + 14].add(b + 13)                          # ... into the running min
        deps[b + 15] |= {b + 0, b + 1, b + 7, b + 9, b + 11}
    cost = {k: (8 if k in o.trans else 5 if k in o.packed else 4) for k in range(n)}
    cost[o.save] = 20
    return n, deps, cost, o.trans


def schedule(nm, xpasses):
    n, deps, cost, trans = build(xpasses)
    succ = {k: set() for k in range(n)}
    for k, ds in deps.items():
        for d in ds:
            succ[d].add(k)
    path = {}

    def longest(k):
        if k not in path:
            path[k] = cost[k] + max((longest(s) for s in succ[k]), default=0)
        return path[k]
    for k in range(n):
        longest(k)
    path[0] = 10 ** 6                       # the dropout-mask load goes out first: its latency is a memory round trip
    total = sum(cost.values())
    ntrans = len(trans)
    max_trans = max(1, -(-ntrans // (nm - nm // 6)))
    done_slot = {}
    order, starts = [], []
    spent = 0
    for s in range(nm):
        starts.append(len(order))
        target = total * (s + 1) / nm
        ready = [k for k in range(n) if k not in done_slot and all(d in done_slot and done_slot[d] < s for d in deps[k])]
        ready.sort(key=lambda k: (-path[k], k))
        picked = []
        nt = 0
        for k in ready:
            if k in trans:
                if nt >= max_trans:
                    continue
                nt += 1
            elif spent + cost[k] > target + 2 and picked:
                continue
            if spent >= target and picked:
                break
            picked.append(k)
            spent += cost[k]
        if s == nm - 1:                     # whatever is left, in dependency order
            rest = [k for k in range(n) if k not in done_slot and k not in picked]
            placed = set(picked)
            while rest:
                for k in list(rest):
                    if all(d in done_slot or d in placed for d in deps[k]):
                        picked.append(k)
                        placed.add(k)
                        rest.remove(k)
        else:
            picked.sort(key=lambda k: (0 if k in trans else 1, -path[k], k))
        for k in picked:
            done_slot[k] = s
        order += picked
    starts.append(len(order))
    assert sorted(order) == list(range(n)), "every operation exactly once"
    pos = {k: i for i, k in enumerate(order)}
    for k, ds in deps.items():
        for d in ds:
            assert pos[d] < pos[k], (d, k)
    worst = max(sum(cost[k] for k in order[starts[s]:starts[s + 1]]) for s in range(nm - 1))
    tr = max(sum(1 for k in order[starts[s]:starts[s + 1]] if k in trans) for s in range(nm))
    return order, starts, worst, tr, len(order) - starts[nm - 1]


def main():
    print("// Generated by tools/gen_lstm_schedule.py — do not edit. Issue order of the gate-math operations of one batch")
    print("// tile and the first position of each MFMA gap's share (see `step` in the kernel header).")
    for nm in (48, 24, 12):
        for with_x in (0, 1):
            order, starts, worst, tr, last = schedule(nm, 2 if with_x else 0)
            tag = f"{nm}{'X' if with_x else ''}"
            print(f"// {nm} gaps, {'with' if with_x else 'without'} the x passes: {len(order)} operations, heaviest gap {worst} issue cycles, "
                  f"<= {tr} transcendentals per gap, {last} operations in the last gap")
            print(f"constexpr int kOrder{tag}[{len(order)}] = {{{', '.join(map(str, order))}}};")
            print(f"constexpr int kStart{tag}[{len(starts)}] = {{{', '.join(map(str, starts))}}};")


if __name__ == "__main__":
    sys.exit(main())
