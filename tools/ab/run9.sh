cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/prof_ml_train
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload movielens-shaped --stages train --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
f=$(find $O -name "*kernel_stats.csv" | head -1); echo $f; head -28 "$f" | cut -c1-170
