#!/usr/bin/env python3
"""HBM-side traffic of the fusion kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of a short full-step run:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -- python3 bench.py --steps 1 --warmup 1 --intervals 16 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/write -- python3 bench.py --steps 1 --warmup 1 --intervals 16 --no-cpu-baseline
    python3 tools/pmc_fusion_traffic.py out/fetch out/write

Same gfx950 corrections as tools/pmc_traffic.py (MI355X_MICROARCH.md, HBM section): read bytes = 2 x FETCH_SIZE x 1024 for
16 B/lane loads, WRITE_SIZE x 1024 for 16 B/lane stores. Per kernel name: mean bytes per dispatch next to the algorithmic
bytes of the default workload (N nodes x T = 16 x d = 64 fp32: the LSTM reads x and writes h, the attention reads h and writes [N, d])."""
import csv
import glob
import json
import sys


def collect(directory, counter, names):
    per = {n: {} for n in names}
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for n in names:
                if n in r["Kernel_Name"]:
                    key = (r["Grid_Size"], r["Dispatch_Id"])
                    per[n][key] = per[n].get(key, 0.0) + float(r["Counter_Value"])
    return per


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    names = ["lstm_fwd_f16_kernel", "ln_mhsa_split_kernel"]
    f, w = collect(fetch_dir, "FETCH_SIZE", names), collect(write_dir, "WRITE_SIZE", names)
    out = {}
    for n in names:
        # users (10 M nodes) and items (5 M) are separate dispatches: report both by their node count (larger traffic = users)
        fr = sorted(v * 2 * 1024 for v in f[n].values())
        wr = sorted(v * 1024 for v in w[n].values())
        half = len(fr) // 2
        for tag, nodes, fs, ws in (("items_5M", 5_000_000, fr[:half], wr[:len(wr) // 2]), ("users_10M", 10_000_000, fr[half:], wr[len(wr) // 2:])):
            if not fs or not ws:
                continue
            rd, wt = sum(fs) / len(fs), sum(ws) / len(ws)
            td = nodes * 16 * 64 * 4
            alg_r, alg_w = td, (td if "lstm" in n else nodes * 64 * 4)
            out[f"{n}/{tag}"] = {"read_bytes": rd, "write_bytes": wt, "algorithmic_read": alg_r, "algorithmic_write": alg_w,
                                 "read_over_algorithmic": rd / alg_r, "write_over_algorithmic": wt / alg_w, "dispatches": len(fs)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
