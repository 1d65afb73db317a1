import json, sys
r = json.load(open(sys.argv[1]))
print("ms/step %.2f" % r["ms_per_step"], "G edges/s %.2f" % (r["value"] / 1e9), "stage", {k: round(v, 2) for k, v in r["stage_ms_per_step_rank0"].items()})
rf = r["roofline"]
print("roofline achieved %.0f GB/s frac %.3f call %.3f ms" % (rf["achieved"], rf["frac"], rf["avg_launch_ms"]), rf.get("kernel_only"), "launches", rf["launches"])
print("by_direction", {k: (round(v["avg_launch_ms"], 3), round(v["frac"], 3)) for k, v in (rf.get("by_direction") or {}).items() if v})
print("redo", r.get("range_redo_tiles_rank0"), "f32 engine", r.get("fusion_f32_engine"))
if "cpu_baseline" in r:
    print("cpu", r["cpu_baseline"]["value"] / 1e6, "M edges/s", r["cpu_baseline"]["cores"], "cores; err", r["cpu_baseline"]["gpu_vs_cpu_max_abs_err"])
    print("item", r.get("item_side_check", {}).get("worst_over_tolerance"), "fused", r.get("fused_check"))
