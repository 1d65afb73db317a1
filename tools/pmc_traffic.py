#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs) of
`bench.py --stages spmm` into profiles/hbm_traffic.json, the `roofline.traffic` source of bench.py.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -- python3 bench.py --steps 2 --warmup 1 --stages spmm --intervals 2 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/write -- python3 bench.py --steps 2 --warmup 1 --stages spmm --intervals 2 --no-cpu-baseline
    python3 tools/pmc_traffic.py out/fetch out/write [--zipf 0.8] [--round 2] > profiles/hbm_traffic.json

The launches of spmm_rows_kernel are split by grid size into the two directions of an interval
(rows = users gathers item rows; rows = items gathers user rows): their cache behaviour differs.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request for
16 B/lane loads, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for 16 B/lane stores."""
import argparse
import csv
import glob
import json
import sys

USERS, ITEMS, NNZ, D = 10_000_000, 5_000_000, 100_000_000, 64


def collect(directory, counter):
    """{grid_size: {dispatch_id: KB}} for spmm_rows_kernel."""
    per = {}
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "spmm_rows_kernel" not in r["Kernel_Name"]:
                continue
            g = per.setdefault(int(r["Grid_Size"]), {})
            g[r["Dispatch_Id"]] = g.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return per


def mean(d):
    return sum(d.values()) / len(d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--zipf", type=float, default=0.8)
    ap.add_argument("--round", type=int, default=2)
    a = ap.parse_args()
    fetch, write = collect(a.fetch_dir, "FETCH_SIZE"), collect(a.write_dir, "WRITE_SIZE")
    # grid = 64 threads x (16-row blocks + long-row chunks): ~29 M threads on the item side (5M rows + hub chunks),
    # ~41 M on the user side (10M rows); the chunk count differs a little between intervals -> cluster at the largest gap
    grids = sorted(fetch)
    if len(grids) < 2 or sorted(write) != grids:
        sys.exit(f"unexpected grid sizes of spmm_rows_kernel: {grids} / {sorted(write)}")
    cut = max(range(1, len(grids)), key=lambda i: grids[i] - grids[i - 1])

    def merged(per, gs):
        out = {}
        for g in gs:
            out.update({f"{g}:{k}": v for k, v in per[g].items()})
        return out
    sides = {}
    for name, gs, rows in (("item_side", grids[:cut], ITEMS), ("user_side", grids[cut:], USERS)):
        g = gs
        fetch_g, write_g = merged(fetch, gs), merged(write, gs)
        rd, wr = 2 * mean(fetch_g) * 1024, mean(write_g) * 1024
        alg = NNZ * (4 * D + 4) + rows * (4 * D + 4 + 4 * D)
        sides[name] = {"grid_sizes": g, "launches": len(fetch_g), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                       "bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg}
    total = (sides["user_side"]["bytes_per_launch"] + sides["item_side"]["bytes_per_launch"]) / 2
    json.dump({
        "workload": "synthetic-powerlaw-10Mx5M", "scale": 1.0, "zipf": a.zipf, "round": a.round,
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py "
                   "--steps 2 --warmup 1 --stages spmm --intervals 2 --no-cpu-baseline"
                   + ("" if a.zipf == 0.8 else f" --zipf {a.zipf:g}") + "; tools/pmc_traffic.py",
        "kernel": "spmm_rows_kernel<16>", "by_direction": sides,
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16 B/lane loads -> read bytes = 2 x FETCH_SIZE x 1024; "
                      "WRITE_SIZE x 1024 is exact for 16 B/lane stores (MI355X_MICROARCH.md, HBM). Counts the L2's memory-side "
                      "requests, Infinity-Cache hits included.",
        "bytes_per_launch": total,
        "algorithmic_bytes_per_launch": (sides["user_side"]["algorithmic_bytes_per_launch"] + sides["item_side"]["algorithmic_bytes_per_launch"]) / 2,
        "plan_thresholds": "short 16 / long 256 / chunk 256",
    }, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
