cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/prof_yelp_train; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 bench.py --workload yelp-shaped --stages train --steps 8 --warmup 2 --no-cpu-baseline > $O/b.json 2> $O/b.err
cp $(find $O/p -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:16]:
    print(r["Name"].replace("void ","").replace("(anonymous namespace)::","")[:64].ljust(64), r["Calls"].rjust(5), ("%.1f us" % (float(r["AverageNs"])/1e3)).rjust(10), ("%.3f ms/step" % (float(r["TotalDurationNs"])/1e6/10)).rjust(14), r["Percentage"])
PY
