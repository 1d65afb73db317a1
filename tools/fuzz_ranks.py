#!/usr/bin/env python3
"""Random (world, intervals, split) combinations of the N > 1 pipeline rehearsed on ONE GPU (gloo transport, real kernels)
against the single-process run of the same intervals: `python tools/fuzz_ranks.py --cases 10 --seed 1`."""
import argparse
import json
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=200)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-1500:])
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')][0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    fails = 0
    for c in range(a.cases):
        world = rnd.choice([2, 3, 4])
        T = rnd.randint(1, 9)
        stages = rnd.choice(["full", "full", "train"])
        split = rnd.choice(["fractional", "groups"]) if T < world else "fractional"
        if stages == "train" and T < world and split == "groups":
            split = "fractional"
        scale = rnd.choice([0.001, 0.002, 0.003])
        common = ["--scale", str(scale), "--no-cpu-baseline", "--intervals", str(T), "--stages", stages,
                  "--steps", "2" if stages == "train" else "1", "--warmup", "0" if stages == "train" else "1"]
        extra = []
        wl = rnd.choice(["synthetic", "synthetic", "gowalla-shaped", "amazon-shaped", "yelp-shaped", "movielens-shaped"])
        if wl != "synthetic":                       # the dataset shapes bring their own T (3 / 5 / 12 / 6) and L; full size is small
            common = ["--workload", wl, "--no-cpu-baseline", "--stages", stages, "--steps", "2" if stages == "train" else "1",
                      "--warmup", "0" if stages == "train" else "1"]
            T = {"gowalla-shaped": 3, "amazon-shaped": 5, "yelp-shaped": 12, "movielens-shaped": 6}[wl]
            split = rnd.choice(["fractional", "groups"]) if (T < world and stages != "train") else "fractional"
        elif T >= world and stages == "full" and rnd.random() < 0.3:
            extra = ["--exchange", "allgather"]
        tag = f"{wl}: world {world}, T {T}, {stages}, split {split}, scale {scale}{' allgather' if extra else ''}"
        print(f"...  {tag}", flush=True)
        try:
            one = run(common)
            many = run(["--gpus", str(world), "--dist-backend", "gloo", "--split", split] + extra + common)
            # dataset shapes run the batched stack at N = 1 and per-interval launches at N > 1: same sums, other association
            tol = 2e-5 if stages == "train" else (1e-6 if (T < world or wl != "synthetic") else 0.0)
            worst = max(abs(x - y) / abs(x) for x, y in zip(one["final_abs_mean"] + one["final_position_checksum"],
                                                            many["final_abs_mean"] + many["final_position_checksum"]))
            ok = worst <= tol
            print(f"{'ok  ' if ok else 'FAIL'} {tag}: worst relative difference {worst:.2e} (allowed {tol:g})", flush=True)
            fails += not ok
        except Exception as e:  # noqa: BLE001
            fails += 1
            print(f"FAIL {tag}: {type(e).__name__}: {str(e)[-600:]}", flush=True)
    print(f"[fuzz_ranks] {a.cases} cases, {fails} failures (seed {a.seed})", flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
