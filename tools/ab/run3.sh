cd $GRAFT_REPO_ROOT
P='import json,sys; r=[json.loads(l) for l in sys.stdin if l.startswith("{\"metric\"")][0]; print(sys.argv[1], r["final_abs_mean"], r["final_position_checksum"], "redo", r["range_redo_tiles_rank0"])'
python bench.py --gpus 4 --dist-backend gloo --steps 1 --warmup 1 --scale 0.002 --no-cpu-baseline 2>gpurun_out/r03_four.err | python -c "$P" N=4
python bench.py --gpus 2 --dist-backend gloo --steps 1 --warmup 1 --scale 0.002 --no-cpu-baseline 2>gpurun_out/r03_two.err | python -c "$P" N=2
python bench.py --gpus 4 --dist-backend gloo --steps 1 --warmup 1 --scale 0.002 --no-cpu-baseline --exchange allgather 2>gpurun_out/r03_four_ag.err | python -c "$P" N=4-allgather
