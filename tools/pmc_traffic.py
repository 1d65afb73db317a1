#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs) of
`bench.py --stages spmm` into profiles/hbm_traffic.json, the `roofline.traffic` source of bench.py.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -- python3 bench.py --steps 2 --warmup 1 --stages spmm --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/write -- python3 bench.py --steps 2 --warmup 1 --stages spmm --no-cpu-baseline
    python3 tools/pmc_traffic.py out/fetch out/write > profiles/hbm_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request for
16 B/lane loads, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for 16 B/lane stores."""
import csv
import glob
import json
import sys


def collect(directory, counter):
    per = {}
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            if "spmm" not in name:
                continue
            per.setdefault(name, {}).setdefault(r["Dispatch_Id"], 0.0)
            per[name][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {"launches": len(v), "mean_KB": sum(v.values()) / len(v), "min_KB": min(v.values()), "max_KB": max(v.values())}
            for k, v in per.items()}


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    rows = next(k for k in fetch if "spmm_rows" in k)
    rd = 2 * fetch[rows]["mean_KB"] * 1024
    wr = write[rows]["mean_KB"] * 1024
    json.dump({
        "workload": "synthetic-powerlaw-10Mx5M", "scale": 1.0, "round": 1,
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py "
                   "--steps 2 --warmup 1 --stages spmm --no-cpu-baseline; tools/pmc_traffic.py",
        "kernel": rows, "counters": {"FETCH_SIZE": fetch, "WRITE_SIZE": write},
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16 B/lane loads -> read bytes = 2 x FETCH_SIZE x 1024; "
                      "WRITE_SIZE x 1024 is exact for 16 B/lane stores (MI355X_MICROARCH.md, HBM). Counts the L2's memory-side "
                      "requests, Infinity-Cache hits included.",
        "bytes_per_launch": rd + wr, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
        "algorithmic_bytes_per_launch": 29870000000.0, "plan_thresholds": "short 16 / long 256 / chunk 256",
    }, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
