cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for wl in gowalla-shaped amazon-shaped movielens-shaped; do
  python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > gpurun_out/r03/bench_${wl}_eager.json
  python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --graph 2>/dev/null | grep '^{"metric"' > gpurun_out/r03/bench_${wl}_graph.json
  python - <<PY
import json
for m in ("eager","graph"):
    r=json.load(open("gpurun_out/r03/bench_${wl}_%s.json"%m))
    print("${wl}", m, "ms/step %.4f"%r["ms_per_step"], "median %.4f"%r["ms_per_step_rank0"]["median"], "stage", {k:round(v,4) for k,v in r["stage_ms_per_step_rank0"].items()}, "launches", r["roofline"]["launches"], "frac %.3f"%r["roofline"]["frac"], "f32 engine", (r.get("fusion_f32_engine") or {}).get("ms_per_step"))
PY
done
