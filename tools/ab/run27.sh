cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/fusion_traffic; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/f.json 2> $O/f.err; echo fetch rc $?
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/w.json 2> $O/w.err; echo write rc $?
python3 tools/pmc_fusion_traffic.py $O/fetch $O/write > $O/fusion_traffic.json; cat $O/fusion_traffic.json
rm -rf $O/fetch $O/write
