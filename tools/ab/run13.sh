cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/prof_ml_train3; rm -rf $O; mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_gpu_fusion_multitile.py tests/test_gpu_backward.py tests/test_gpu_configs.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
grep -q passed $O/tests.log && ! grep -q failed $O/tests.log || exit 1
python bench.py --workload movielens-shaped --stages train --steps 8 --warmup 2 --no-cpu-baseline > $O/b_plain.json 2> $O/b_plain.err
python tools/ab/show.py $O/b_plain.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 bench.py --workload movielens-shaped --stages train --steps 8 --warmup 2 --no-cpu-baseline > $O/b.json 2> $O/b.err
cp $(find $O/p -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
head -14 $O/kernel_stats.csv | cut -c1-150
