cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/traffic; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --stages spmm --intervals 2 --no-cpu-baseline > $O/f.json 2> $O/f.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --stages spmm --intervals 2 --no-cpu-baseline > $O/w.json 2> $O/w.err
python3 tools/pmc_traffic.py $O/fetch $O/write --round 3 > $O/r03_hbm_traffic.json 2> $O/t.err; cat $O/r03_hbm_traffic.json | head -40; cat $O/t.err | tail -3
rm -rf $O/fetch $O/write
