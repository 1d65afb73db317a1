cd $GRAFT_REPO_ROOT
run() { python3 -c "
import sys, runpy
sys.path.insert(0, '.')
import sa_gnn_amd.autograd as ag
ag.SPLIT_DW = $1
sys.argv = ['bench.py'] + '$2'.split()
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][0]); print('split=$1', '$2'.split()[1], round(r['ms_per_step'],3), 'redo', r['range_redo_tiles_rank0'])"; }
for wl in yelp-shaped amazon-shaped; do
  for m in False True False True; do run $m "--workload $wl --stages train --steps 10 --warmup 2 --no-cpu-baseline"; done
done
for m in False True; do run $m "--workload synthetic-powerlaw-10Mx5M --stages train --scaling weak --intervals-per-gpu 4 --scale 0.25 --steps 4 --warmup 1 --no-cpu-baseline"; done
