# Round-3 evidence run (1 x MI355X). Writes under gpurun_out/r03/final/; the summaries are copied to profiles/ by hand.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/final
rm -rf $O; mkdir -p $O
st() { echo "$(date +%T) $*" >> $O/status.txt; }
st start
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > $O/bench_default_n1.json 2> $O/bench_default_n1.err; st "default rc $?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_default_under_rocprof_n1.json 2> $O/prof_default.err; st "rocprof default rc $?"
timeout -k 10 300 python3 bench.py --stages train --scaling weak --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_n1.json 2> $O/bench_train.err; st "train rc $?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 bench.py --stages train --scaling weak --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_train_under_rocprof.json 2> $O/prof_train.err; st "rocprof train rc $?"
for wl in gowalla-shaped amazon-shaped movielens-shaped; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -- python3 bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_${wl}_under_rocprof.json 2> $O/prof_$wl.err; st "rocprof $wl rc $?"
  timeout -k 10 200 python3 bench.py --workload $wl --stages train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${wl}_train.json 2> $O/train_$wl.err; st "train $wl rc $?"
done
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq.json 2> $O/pmc_sq.err; st "pmc rc $?"
python3 tools/pmc_sq.py $O/pmc_sq spmm_rows lstm_fwd ln_mhsa > $O/pmc_sq_summary.txt 2>&1
for d in prof_default prof_train prof_gowalla-shaped prof_amazon-shaped prof_movielens-shaped; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_${d#prof_}.csv; done
rm -rf $O/pmc_sq $O/prof_*/
st done
cat $O/status.txt
timeout -k 10 200 python3 bench.py --workload yelp-shaped --stages train --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_yelp-shaped_train.json 2> $O/train_yelp.err; st "train yelp rc $?"
timeout -k 10 200 python3 bench.py --workload yelp-shaped --steps 20 --warmup 3 --graph --no-cpu-baseline > $O/bench_yelp-shaped_graph.json 2> $O/graph_yelp.err; st "graph yelp rc $?"
for wl in gowalla-shaped amazon-shaped movielens-shaped; do
  timeout -k 10 200 python3 bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_${wl}_eager.json 2> $O/eager_$wl.err; st "eager $wl rc $?"
  timeout -k 10 200 python3 bench.py --workload $wl --steps 50 --warmup 5 --graph --no-cpu-baseline > $O/bench_${wl}_graph.json 2> $O/graph_$wl.err; st "graph $wl rc $?"
done
cat $O/status.txt
