cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python bench.py --steps 10 --warmup 3 2>gpurun_out/r03/bench_default.err | grep '^{"metric"' > gpurun_out/r03/bench_default.json
python tools/ab/show.py gpurun_out/r03/bench_default.json
