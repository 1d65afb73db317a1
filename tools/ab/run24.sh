cd $GRAFT_REPO_ROOT
for wl in amazon-shaped yelp-shaped gowalla-shaped; do
python3 bench.py --workload $wl --stages train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][0]); print('$wl', round(r['ms_per_step'],3), 'redo', r['range_redo_tiles_rank0'], 'final', r['final_abs_mean'])"
done
timeout -k 10 900 python -m pytest tests/test_gpu_f16_range.py tests/test_gpu_dense.py -x -q 2>&1 | tail -3
