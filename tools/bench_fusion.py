#!/usr/bin/env python3
"""Times the interval-fusion kernels alone (LSTM, LN + MHSA + mean) at the roofline configuration's
row count, for the three GEMM engines: the default (f16 matrix cores, operands split in two round-to-nearest
pieces, three piece products), and the exact-fp32 engine (v_mfma_f32_32x32x2_f32; ops.set_engine('f32')), and prints how far the results are apart and how far each is from
the numpy oracle on a row slice.  python tools/bench_fusion.py [--n 15000000] [--t 2] [--d 64]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=15_000_000)
    ap.add_argument("--t", type=int, default=2)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--engines", default="f16x2,bf16x3,f32", help="comma-separated subset of f16x2, bf16x3, f32")
    ap.add_argument("--train", action="store_true", help="also time the training forward (stores gates / cell)")
    a = ap.parse_args()
    from oracle import selfgnn_oracle as O
    from sa_gnn_amd import ops
    from sa_gnn_amd.model import random_fusion_params
    dev = torch.device("cuda:0")
    n, t, d = a.n, a.t, a.d
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand((t, n, d), generator=g, device=dev).mul_(2).sub_(1).permute(1, 0, 2)     # [n, t, d] view of [t, n, d]
    p = random_fusion_params(d, dev, 7)
    h = torch.empty((n, t, d), device=dev)
    S = min(n, 50_000)
    pn = {k: v.cpu().numpy() for k, v in p.items()}
    xs = np.ascontiguousarray(x[:S].cpu().numpy())
    want_h = O.basic_lstm(xs, pn["lstm_W"], pn["lstm_b"], 1.0)
    want_f = O.mhsa(O.layer_norm_td(want_h, pn["ln_gamma"], pn["ln_beta"]), pn["Wq"], pn["bq"], pn["Wk"], pn["bk"], pn["Wv"],
                    pn["bv"], 16).mean(axis=1)
    res = {}
    engines = a.engines.split(",")
    for mode in engines:
        ops.set_engine(mode)

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ts)), float(np.min(ts))
        lstm_ms = timed(lambda: ops.lstm_fwd(x, p["lstm_W"], p["lstm_b"], out=h))
        hh = h[:S].cpu().numpy()
        fused = [None]

        def attn():
            fused[0] = ops.ln_mhsa_mean(h, p["ln_gamma"], p["ln_beta"], p["Wq"], p["bq"], p["Wk"], p["bk"], p["Wv"], p["bv"], 16)
        attn_ms = timed(attn)
        ff = fused[0][:S].cpu().numpy()
        res[mode] = (hh, ff, h[:, -1, :].double().abs().mean().item(), fused[0].double().abs().mean().item())
        flop_lstm = n * (16 * d * d * t - 8 * d * d)
        flop_attn = n * t * 6 * d * d
        print(f"[{mode:6s}] n={n} t={t} d={d}: LSTM {lstm_ms[0]:.3f} ms (min {lstm_ms[1]:.3f}) = {flop_lstm / lstm_ms[0] / 1e9:.1f} TFLOP/s fp32-equivalent; "
              f"LN+MHSA {attn_ms[0]:.3f} ms (min {attn_ms[1]:.3f}) = {flop_attn / attn_ms[0] / 1e9:.1f} TFLOP/s; "
              f"vs oracle (first {S} rows): h max abs err {np.abs(hh - want_h).max():.3e}, fused {np.abs(ff - want_f).max():.3e}", flush=True)
        if a.train:
            gates = torch.empty((n, t, 4 * d), device=dev)
            cell = torch.empty((n, t, d), device=dev)
            lib = ops._lib.load()

            def train_fwd():
                ops.check(lib.sagnn_lstm_fwd_train_f32(x.data_ptr(), x.stride(0), x.stride(1), n, t, d, p["lstm_W"].data_ptr(),
                                                       p["lstm_b"].data_ptr(), 1.0, None, h.data_ptr(), t * d, gates.data_ptr(),
                                                       cell.data_ptr(), ops._stream()))
            tr = timed(train_fwd)
            print(f"[{mode:6s}] training forward (stores gates + cell): {tr[0]:.3f} ms", flush=True)
            del gates, cell
    for mode in [m for m in engines if m != "f32" and "f32" in engines]:
        dh = np.abs(res[mode][0] - res["f32"][0]).max()
        df = np.abs(res[mode][1] - res["f32"][1]).max()
        print(f"{mode} vs f32 engines: h max abs diff {dh:.3e}, fused max abs diff {df:.3e}; "
              f"abs-mean of last h {res[mode][2]:.9f} / {res['f32'][2]:.9f}, of fused {res[mode][3]:.9f} / {res['f32'][3]:.9f}")


if __name__ == "__main__":
    main()
