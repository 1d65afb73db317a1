#!/usr/bin/env python3
"""Randomised sweep of the GPU parity tests: calls the test functions of tests/test_gpu_*.py with parameters drawn at random
(shapes the parametrised lists do not contain) for a fixed time budget, prints every failure with its parameters.

    python tools/fuzz_gpu.py --seconds 300 --seed 1
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    import torch
    import test_gpu_backward as tb
    import test_gpu_dense as td
    import test_gpu_f16_range as tr
    import test_gpu_fusion as tf
    import test_gpu_fusion_multitile as tm
    dev = torch.device("cuda:0")
    rnd = random.Random(a.seed)
    T_FAST = [1, 2, 3, 4, 5, 6, 8, 12, 16]

    def n_(lo=1, hi=5000):
        return rnd.choice([rnd.randint(lo, max(lo, 40)), rnd.randint(lo, hi), max(lo, rnd.choice([31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 191, 193]))])

    cases = [
        ("lstm_vs_oracle", lambda: tf.test_lstm_vs_oracle(dev, rnd.choice([32, 64, 128]), rnd.choice(T_FAST + [7, 9]), n_())),
        ("mhsa_mean_vs_oracle", lambda: tf.test_mhsa_mean_vs_oracle(dev, rnd.choice([32, 64, 128]), 16, rnd.choice(T_FAST + [7, 10]), n_(1, 2000))),
        ("interval_fusion_vs_oracle", lambda: tf.test_interval_fusion_vs_oracle(dev, rnd.choice([32, 64, 128]), rnd.choice(T_FAST), n_(1, 3000))),
        ("position_independence", lambda: tf.test_a_rows_result_does_not_depend_on_its_position_in_a_tile(dev, rnd.choice([32, 64]), rnd.choice(T_FAST))),
        ("fusion_backward", lambda: tb.test_interval_fusion_backward_vs_autograd(dev, rnd.choice([32, 64, 128]), rnd.choice([1, 2, 3, 4, 5, 6]), n_(1, 400), 16)),
        ("fusion_backward_longT", lambda: tb.test_interval_fusion_backward_vs_autograd(dev, rnd.choice([32, 64]), rnd.choice([8, 12, 16]), n_(1, 200), 16)),
        ("dense_nn", lambda: td.test_dense_nn(dev, n_(1, 6000), 32 * rnd.randint(1, 16), 32 * rnd.randint(1, 16))),
        ("dense_tn", lambda: td.test_dense_tn(dev, n_(1, 20000), 32 * rnd.randint(1, 12), 32 * rnd.randint(1, 16))),
        ("dense_tn_seg", lambda: td.test_dense_tn_seg(dev, n_(1, 3000), rnd.randint(1, 12), 32 * rnd.randint(1, 6), 32 * rnd.randint(1, 16),
                                                     rnd.choice(["node", "time"]))),
        ("attn_bwd_tail", lambda: td.test_attn_bwd_tail(dev, n_(1, 50000), rnd.choice([32, 64]), rnd.choice(["f16x2", "f32"]))),
        ("tail_any_scale", lambda: tr.test_attn_bwd_tail_rows_of_any_scale(dev, n_(33, 30000), rnd.choice([32, 64]),
                                                                           rnd.choice(["blocks", "mixed", "rising", "falling", "tiny", "needle"]), "f16x2")),
        ("lstm_dw_any_scale", lambda: tr.test_lstm_weight_gradient_pass_at_any_gradient_scale(dev, rnd.choice([32, 64]), rnd.randint(1, 9), n_(1, 9000),
                                                                                              rnd.choice(["blocks", "mixed", "tiny"]))),
        ("lstm_dw_dropout", lambda: tr.test_lstm_weight_gradient_pass_with_an_output_dropout_mask(dev, rnd.choice([32, 64]), rnd.randint(1, 8), n_(1, 5000))),
        ("attn_bwd_front", lambda: tm.test_attn_bwd_front_many_tiles_per_block(dev, *rnd.choice([(64, rnd.choice(T_FAST)), (32, rnd.choice(T_FAST)),
                                                                                 (128, rnd.choice([1, 2, 3, 4, 5, 6]))]), n_(1, 6000))),
    ]
    import functools

    def traced(mod):
        """Wrap every test function of a module so that a failure can name the arguments it was called with."""
        for nm in dir(mod):
            f = getattr(mod, nm)
            if nm.startswith("test_") and callable(f):
                def w(*args, _f=f, _nm=nm, **kw):
                    last["call"] = f"{_nm}{args[1:]}"
                    return _f(*args, **kw)
                setattr(mod, nm, functools.wraps(f)(w))

    last = {"call": ""}
    for m in (tb, td, tr, tf, tm):
        traced(m)
    def spmm_case():
        """Random interval graphs (hub rows / columns, empty rows, an empty interval now and then), random plan thresholds, any
        d the SpMM takes; the per-interval entry and the batched entry against the oracle and against each other."""
        import numpy as np
        import scipy.sparse as sp
        import test_gpu_spmm as ts
        from oracle import selfgnn_oracle as O
        from sa_gnn_amd import graph, ops
        rng = np.random.default_rng(rnd.randrange(1 << 30))
        U, I, T, L = rnd.randint(1, 400), rnd.randint(1, 300), rnd.randint(1, 5), rnd.randint(1, 3)
        d = rnd.choice([4, 8, 16, 32, 48, 64, 96, 128, 256])
        short, long_, chunk = rnd.choice([(2, 4, 32), (4, 8, 64), (8, 24, 64), (16, 256, 256), (8, 16, 32)])
        last["call"] = f"spmm_case(U={U}, I={I}, T={T}, L={L}, d={d}, tuning={(short, long_, chunk)})"
        mats = []
        for k in range(T):
            m = (rng.random((U, I)) < rnd.choice([0.0, 0.01, 0.05, 0.2])).astype(np.intc)
            if rnd.random() < 0.5:
                m[rnd.randrange(U), :] = 1
            if rnd.random() < 0.5:
                m[:, rnd.randrange(I)] = 1
            if rnd.random() < 0.3 and U > 3:
                m[U // 2:, :] = 0
            mats.append(sp.csr_matrix(m))
        pairs = [graph.interval_pair(m, dev, tuning=(short, long_, chunk)) for m in mats]
        ue = rng.standard_normal((T, U, d)).astype(np.float32)
        ie = rng.standard_normal((T, I, d)).astype(np.float32)
        ued, ied = torch.from_numpy(ue).to(dev), torch.from_numpy(ie).to(dev)
        adjs = [O.trans_to_lsts(m)[0] for m in mats]
        tps = [O.trans_to_lsts(O.transpose(m))[0] for m in mats]
        want_u, want_i = O.gnn_stack(ue, ie, adjs, tps, L, 0.5)
        terms_u, terms_i = O.gnn_stack(np.abs(ue), np.abs(ie), adjs, tps, L, 1.0)
        us, its = torch.empty((U, T, d), device=dev), torch.empty((I, T, d), device=dev)
        for k in range(T):
            ops.gnn_interval(pairs[k][0].plan, pairs[k][1].plan, ued[k], ied[k], L, 0.5, us[:, k, :], its[:, k, :])
        ts.assert_sum_close(us.cpu().numpy(), want_u, terms_u)
        ts.assert_sum_close(its.cpu().numpy(), want_i, terms_i)
        batch = ops.SpmmBatch([p_[0].plan for p_ in pairs], [p_[1].plan for p_ in pairs])
        us2, its2 = torch.empty((U, T, d), device=dev), torch.empty((I, T, d), device=dev)
        ops.gnn_stack(batch, ued, ied, L, 0.5, us2.permute(1, 0, 2), its2.permute(1, 0, 2))
        assert torch.equal(us, us2) and torch.equal(its, its2), "batched entry differs from the per-interval entry"

    cases.append(("spmm", spmm_case))
    cases.append(("spmm", spmm_case))            # the path's own kernel: drawn twice as often
    eng = lambda: rnd.choice(["f16x2", "f16x2", "f32"])   # noqa: E731
    cases += [
        ("dx_any_scale", lambda: tr.test_fusion_backward_dx_per_node_at_any_gradient_scale(dev, rnd.choice([32, 64]), rnd.choice([1, 2, 3, 4, 5, 6]), n_(40, 4000), eng())),
        ("lstm_small_inputs", lambda: tr.test_lstm_small_inputs_keep_their_own_accuracy(dev, rnd.choice([32, 64, 128]), rnd.choice(T_FAST[2:]), n_(300, 3000), eng())),
        ("attention_small_inputs", lambda: tr.test_attention_small_inputs_keep_their_own_accuracy(dev, rnd.choice([32, 64, 128]), rnd.choice(T_FAST), n_(300, 3000), eng())),
        ("tail_beyond_range", lambda: td.test_attn_bwd_tail_beyond_the_f16_range(dev, n_(40, 30000), rnd.choice([32, 64]))),
        ("lstm_beyond_range", lambda: tm.test_lstm_inputs_beyond_the_f16_range(dev, rnd.choice([32, 64, 128]), rnd.choice(T_FAST[1:]), n_(700, 5000))),
        ("attention_beyond_range", lambda: tm.test_attention_operands_beyond_the_f16_range(dev, rnd.choice([32, 64, 128]), rnd.choice(T_FAST[1:]), n_(300, 5000))),
        ("lstm_continuation", lambda: tf.test_lstm_continuation_is_bit_identical(dev, rnd.choice([32, 64, 128]), n_(1, 3000))),
        ("engines_agree", lambda: tm.test_split_engine_matches_f32_mfma_engine(dev, *rnd.choice([(64, rnd.choice(T_FAST)), (32, rnd.choice(T_FAST)),
                                                                                  (128, rnd.choice([1, 2, 3, 4, 5, 6]))]), n_(1, 6000))),
        # (T >= 2: with one key the attention weights are 1 whatever q and k are — dWq / dWk are fp32 noise around ~0 and the test's noise floor is a rule of thumb)
        ("training_fwd_bwd", lambda: tm.test_training_forward_and_backward_many_tiles_per_block(dev, rnd.choice([32, 64]), rnd.choice(T_FAST[1:]), n_(1, 3000))),
    ]
    t_end = time.time() + a.seconds
    runs = fails = 0
    counts = {}
    while time.time() < t_end:
        name, fn = rnd.choice(cases)
        counts[name] = counts.get(name, 0) + 1
        last["call"] = name
        try:
            fn()
        except KeyboardInterrupt:
            raise
        except BaseException as e:  # noqa: BLE001  (pytest.skip raises outside Exception)
            if type(e).__name__ == "Skipped":
                continue
            fails += 1
            print(f"FAIL {last['call']}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        runs += 1
        if runs % 25 == 0:
            print(f"[fuzz] {runs} runs, {fails} failures", flush=True)
    print(f"[fuzz] done: {runs} runs, {fails} failures (seed {a.seed}); per case: {dict(sorted(counts.items()))}", flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
