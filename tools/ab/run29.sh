cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/overlap; rm -rf $O; mkdir -p $O
timeout -k 10 300 python bench.py --scale 0.01 --steps 2 --warmup 1 --no-cpu-baseline > $O/small.json 2> $O/small.err; echo small rc $?; tail -3 $O/small.err
python - <<PY
import json
r=json.load(open("$O/small.json"))
print(r["ms_per_step"], r["config"]["schedule"][:40], r.get("serial_schedule"), r["roofline"].get("overlapped_launches"))
PY
timeout -k 10 500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/full.json 2> $O/full.err; echo full rc $?; tail -2 $O/full.err
python tools/ab/show.py $O/full.json
python - <<PY
import json
r=json.load(open("$O/full.json"))
print("serial", r.get("serial_schedule"))
print("overlapped", r["roofline"].get("overlapped_launches"))
PY
for m in 4 12 16; do timeout -k 10 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --overlap-launches $m 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][0]); print('m=$m', round(r['ms_per_step'],2), 'serial', round(r['serial_schedule']['ms_per_step'],2), r['roofline']['frac'], r['roofline'].get('overlapped_launches',{}).get('avg_launch_ms'))"; done
