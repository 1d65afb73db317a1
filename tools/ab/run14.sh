cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/rccl1; rm -rf $O; mkdir -p $O
timeout -k 10 300 python bench.py --dist-single --steps 1 --warmup 1 --scale 0.004 --no-cpu-baseline --intervals 4 > $O/full.json 2> $O/full.err; echo rc $?; tail -5 $O/full.err; cat $O/full.json | cut -c1-600
timeout -k 10 300 python bench.py --dist-single --steps 2 --warmup 0 --scale 0.004 --no-cpu-baseline --intervals 4 --stages train > $O/train.json 2> $O/train.err; echo rc $?; tail -5 $O/train.err; cat $O/train.json | cut -c1-300
timeout -k 10 300 python bench.py --dist-single --exchange allgather --steps 1 --warmup 1 --scale 0.004 --no-cpu-baseline --intervals 4 > $O/ag.json 2> $O/ag.err; echo rc $?; tail -5 $O/ag.err; cat $O/ag.json | cut -c1-300
