#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 SQ counter pass, with the ratios the kernel work is steered by.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \\
              SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d out/sq -- python3 <cmd>
    python3 tools/pmc_sq.py out/sq [name-filter ...]

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES counts cycles of the
matrix pipe per SIMD (MI355X_MICROARCH.md, cycle constants), so MFMA busy is priced against 4 SIMDs x CU-busy cycles."""
import csv
import glob
import sys


def main():
    directory, filters = sys.argv[1], sys.argv[2:]
    per = {}
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            if filters and not any(s in name for s in filters):
                continue
            k = per.setdefault(name, {"_d": set()})
            k["_d"].add(r["Dispatch_Id"])
            k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for name, k in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
        g = lambda c: k.get(c, 0.0)       # noqa: E731
        wc = g("SQ_WAVE_CYCLES") or 1.0
        print(f"{name}: dispatches {len(k['_d'])}")
        print("    " + "  ".join(f"{c}={v:.3e}" for c, v in sorted(k.items()) if c != "_d"))
        print(f"    of wave cycles: active {g('SQ_ACTIVE_INST_ANY') / wc:.3f}, parked on waitcnt/barrier {g('SQ_WAIT_ANY') / wc:.3f}, "
              f"issue-stalled {g('SQ_WAIT_INST_ANY') / wc:.3f}; MFMA busy / (4 SIMD x CU busy) "
              f"{g('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * (g('SQ_BUSY_CU_CYCLES') or 1.0)):.3f}; LDS conflict / LDS active "
              f"{g('SQ_LDS_BANK_CONFLICT') / (g('SQ_LDS_IDX_ACTIVE') or 1.0):.3f}")


if __name__ == "__main__":
    main()
