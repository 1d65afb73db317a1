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
    def __init__(self, u):
        self.u = u
        self.c, self.e, self.a, self.r = 1, 1 + 4 * u, 1 + 8 * u, 1 + 12 * u
        t0 = 1 + 16 * u
        (self.tj, self.pr, self.cn, self.um, self.ue, self.ua, self.ur, self.ut, self.hn, self.hv) = (t0 + i * u for i in range(10))
        self.head = t0 + 10 * u
        self.res = self.head + u // 2
        self.tail = self.res + u
        self.write = self.tail + u
        self.store = self.write + 1
        self.save = self.write + 2
        self.n_gate = self.write + 3
        self.trans = (set(range(self.e, self.e + 4 * u)) | set(range(self.r, self.r + 4 * u)) |
                      set(range(self.ue, self.ue + u)) | set(range(self.ur, self.ur + u)))


def build(u, xpasses):
    o = Ops(u)
    n = o.n_gate + xpasses * K_X
    deps = {k: set() for k in range(n)}
    for k in range(4 * u):
        deps[o.e + k].add(o.c + k)
        deps[o.a + k].add(o.e + k)
        deps[o.r + k].add(o.a + k)
    for i in range(u):
        deps[o.tj + i].add(o.r + 1 * u + i)
        deps[o.pr + i] |= {o.r + i, o.tj + i}
        deps[o.cn + i] |= {o.r + 2 * u + i, o.pr + i}
        deps[o.um + i].add(o.cn + i)
        deps[o.ue + i].add(o.um + i)
        deps[o.ua + i].add(o.ue + i)
        deps[o.ur + i].add(o.ua + i)
        deps[o.ut + i].add(o.ur + i)
        deps[o.hn + i] |= {o.ut + i, o.r + 3 * u + i}
        deps[o.hv + i] |= {o.hn + i, 0}
        deps[o.res + i] |= {o.head + i // 2, o.hn + i}
        deps[o.tail + i].add(o.res + i)
        if i % 2:
            deps[o.tail + i].add(o.tail + i - 1)          # v_fma_mixhi_f16 into the register v_fma_mixlo_f16 wrote
    for j in range(u // 2):
        deps[o.head + j] |= {o.hn + 2 * j, o.hn + 2 * j + 1}
    deps[o.write] |= set(range(o.head, o.head + u // 2)) | set(range(o.tail, o.tail + u))
    deps[o.store] |= set(range(o.hv, o.hv + u)) | set(range(o.cn, o.cn + u))
    deps[o.save] |= set(range(o.r, o.r + 4 * u)) | set(range(o.tj, o.tj + u)) | set(range(o.cn, o.cn + u))
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
    cost = {k: (8 if k in o.trans else 4) for k in range(n)}
    cost[o.save] = 20
    return n, deps, cost, o.trans


def schedule(u, nm, xpasses):
    n, deps, cost, trans = build(u, xpasses)
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
    u = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    xp = 2 if u == 4 else 1
    gaps = (48, 24, 12) if u == 4 else (24, 12, 6)
    print(f"// Generated by tools/gen_lstm_schedule.py {u} — do not edit. Issue order of the gate-math operations of one batch")
    print("// tile and the first position of each MFMA gap's share (see `step` in the kernel header).")
    for nm in gaps:
        for with_x in (0, 1):
            order, starts, worst, tr, last = schedule(u, nm, xp if with_x else 0)
            tag = f"{nm}{'X' if with_x else ''}"
            print(f"// {nm} gaps, {'with' if with_x else 'without'} the x pass{'es' if xp > 1 else ''}: {len(order)} operations, heaviest gap {worst} issue cycles, "
                  f"<= {tr} transcendentals per gap, {last} operations in the last gap")
            print(f"constexpr int kOrder{tag}[{len(order)}] = {{{', '.join(map(str, order))}}};")
            print(f"constexpr int kStart{tag}[{len(starts)}] = {{{', '.join(map(str, starts))}}};")


if __name__ == "__main__":
    sys.exit(main())
