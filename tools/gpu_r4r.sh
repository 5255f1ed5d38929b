#!/bin/bash
# round 4, sessions r, r2: row-of-16 partial sums (a), dfp by readlane (d), publish merged into the edge phase (m): singly (r) and combined (r2)
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'loop ms', round(d['ms_per_step'],4), 'kernel ms', round(r.get('avg_launch_us', 0)/1e3,4), 'frac', round(r['frac'],4), d.get('parity',{}).get('rule','')[:60])" 2>&1 | tail -1; }
for rep in 1 2; do
for v in a ad am adm; do
  GNNX_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_build/ab/libgnnx_hip_$v.so timeout 200 python bench.py --loop-only --steps 20 --warmup 5 --no-cpu-baseline --reps 1 > $O/loop_${v}_$rep.json 2> $O/loop_${v}_$rep.err; show $O/loop_${v}_$rep.json "syn1 $v run $rep"
done; done
for v in a adm; do
  GNNX_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_build/ab/libgnnx_hip_$v.so timeout 200 python bench.py --workload syn5 --loop-only --steps 20 --warmup 5 --no-cpu-baseline --reps 1 > $O/loop_syn5_$v.json 2> /dev/null; show $O/loop_syn5_$v.json "syn5 $v"
done
