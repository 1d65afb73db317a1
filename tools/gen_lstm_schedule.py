#!/usr/bin/env python3
"""Generates sa-gnn_amd/csrc/lstm_f16_schedule.inc: the order in which the gate-math operations of one
16-row batch tile (numbered as in gate_op of lstm_f16_kernel.h) are issued between the MFMAs of the next
tile, and where each MFMA slot's share begins.

A list scheduler over the operations' dependency graph: every MFMA gap gets at most `max_trans`
transcendentals (v_exp_f32 / v_rcp_f32: 8 issue cycles, the rest 4 — MI355X_MICROARCH.md, per-instruction
constants) and plain operations up to an even share of the total issue cost; an operation is ready
when everything it reads was issued in an EARLIER gap (>= 16 cycles before: no result latency is exposed).
Critical-path-first. Six tables: gaps per tile NM = 48 / 24 / 12 (d = 64 recurrent step / its first step and d = 32 / d = 32 first step), each with and
without the 2 x 13 operations that split two passes of the next step's x.

    python tools/gen_lstm_schedule.py > sa-gnn_amd/csrc/lstm_f16_schedule.inc"""
import sys

K_GATE, K_X = 118, 13
TRANS = set(range(17, 33)) | set(range(49, 65)) | set(range(81, 85)) | set(range(89, 93))


def build(with_x):
    n = K_GATE + (2 * K_X if with_x else 0)
    deps = {k: set() for k in range(n)}
    for k in range(16):
        deps[17 + k].add(1 + k)          # exp2 <- joined accumulators
        deps[33 + k].add(17 + k)         # 1 + e
        deps[49 + k].add(33 + k)         # rcp
    for r in range(4):
        deps[65 + r].add(49 + 4 + r)                     # tanh(j)
        deps[69 + r] |= {49 + r, 65 + r}                 # i * tanh(j)
        deps[73 + r] |= {49 + 8 + r, 69 + r}             # c' = c f + .
        deps[77 + r].add(73 + r)
        deps[81 + r].add(77 + r)
        deps[85 + r].add(81 + r)
        deps[89 + r].add(85 + r)
        deps[93 + r].add(89 + r)
        deps[97 + r] |= {93 + r, 49 + 12 + r}            # h = tanh(c') o
        deps[101 + r] |= {97 + r, 0}                     # dropout scale
        deps[107 + r] |= {105 + r // 2, 97 + r}          # residual of h
    deps[105] |= {97, 98}
    deps[106] |= {99, 100}
    deps[111].add(107)
    deps[112] |= {111, 108}
    deps[113].add(109)
    deps[114] |= {113, 110}
    deps[115] |= {105, 106, 112, 114}
    deps[116] |= {101, 102, 103, 104, 73, 74, 75, 76}
    deps[117] |= set(range(49, 65)) | {65, 66, 67, 68, 73, 74, 75, 76}
    if with_x:
        for p in range(2):
            b = K_GATE + p * K_X
            for i in range(4):
                deps[b + 2 + i].add(b + i // 2)
            deps[b + 6].add(b + 2)
            deps[b + 7] |= {b + 6, b + 3}
            deps[b + 8].add(b + 4)
            deps[b + 9] |= {b + 8, b + 5}
            deps[b + 12] |= {b + 0, b + 1, b + 7, b + 9, b + 11}
    cost = {k: (8 if k in TRANS else 4) for k in range(n)}
    cost[117] = 20
    return n, deps, cost


def schedule(nm, with_x):
    n, deps, cost = build(with_x)
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
    max_trans = 1 if nm >= 40 else 2 if nm >= 20 else 4
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
            if k in TRANS:
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
        picked.sort(key=lambda k: (0 if k in TRANS else 1, -path[k], k)) if s < nm - 1 else None
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
    tr = max(sum(1 for k in order[starts[s]:starts[s + 1]] if k in TRANS) for s in range(nm))
    return order, starts, worst, tr, len(order) - starts[nm - 1]


def main():
    print("// Generated by tools/gen_lstm_schedule.py — do not edit. Issue order of the gate-math operations of one batch")
    print("// tile and the first position of each MFMA gap's share (see lstm_f16_kernel.h, `step`).")
    for nm in (48, 24, 12):
        for with_x in (0, 1):
            order, starts, worst, tr, last = schedule(nm, with_x)
            tag = f"{nm}{'X' if with_x else ''}"
            print(f"// {nm} gaps, {'with' if with_x else 'without'} the x passes: {len(order)} operations, heaviest gap {worst} issue cycles, "
                  f"<= {tr} transcendentals per gap, {last} operations in the last gap")
            print(f"constexpr int kOrder{tag}[{len(order)}] = {{{', '.join(map(str, order))}}};")
            print(f"constexpr int kStart{tag}[{len(starts)}] = {{{', '.join(map(str, starts))}}};")


if __name__ == "__main__":
    sys.exit(main())
