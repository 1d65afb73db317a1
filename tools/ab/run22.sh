cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/bptt_split; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 tools/ab/bptt_split.py > $O/out.txt 2> $O/err.txt
cp $(find $O/p -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:8]:
    print(r["Name"].replace("void ","").replace("(anonymous namespace)::","")[:70].ljust(70), r["Calls"].rjust(5), ("%.1f us" % (float(r["AverageNs"])/1e3)).rjust(12), r["Percentage"])
PY
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/p/*/*kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    nm=r["Kernel_Name"]
    if "lstm_bwd" in nm or "tail_f16" in nm:
        d[(nm[:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for k,v in d.items(): print(k, len(v), "%.2f ms" % (sum(v)/len(v)))
PY
