cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/rccl1; mkdir -p $O
timeout -k 10 400 python bench.py --dist-single --scale 0.25 --steps 5 --warmup 2 --no-cpu-baseline > $O/rccl_single_rank_scale025.json 2> $O/a.err; echo rc $?; tail -2 $O/a.err
timeout -k 10 400 python bench.py --scale 0.25 --steps 5 --warmup 2 --no-cpu-baseline > $O/plain_scale025.json 2> $O/b.err; echo rc $?
python - <<PY
import json
a=json.load(open("$O/rccl_single_rank_scale025.json")); b=json.load(open("$O/plain_scale025.json"))
print("dist-single", a["ms_per_step"], a.get("breakdown_ms"), a["final_abs_mean"], a["final_position_checksum"])
print("plain      ", b["ms_per_step"], b["final_abs_mean"], b["final_position_checksum"])
PY
head -c 60 $O/rccl_single_rank_scale025.json; echo; wc -l $O/rccl_single_rank_scale025.json
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
