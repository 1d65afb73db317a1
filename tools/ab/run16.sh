cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/yelp; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -k yelp > $O/tests.log 2>&1; tail -5 $O/tests.log
python bench.py --workload yelp-shaped --steps 20 --warmup 3 --graph --no-cpu-baseline > $O/graph.json 2> $O/graph.err; python tools/ab/show.py $O/graph.json
python bench.py --workload yelp-shaped --steps 10 --warmup 2 --stages train --no-cpu-baseline > $O/train.json 2> $O/train.err; python tools/ab/show.py $O/train.json
