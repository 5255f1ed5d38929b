#!/bin/bash
# round 4, session o: per-row slot width for the rows of t and its neighbours (b), relu(U1) stored (r), on top of fmac + SGPR wsplit (fs0)
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'loop ms', round(d['ms_per_step'],4), 'kernel ms', round(r.get('avg_launch_us', 0)/1e3,4), 'frac', round(r['frac'],4), d.get('parity',{}).get('rule','')[:60])" 2>&1 | tail -1; }
for rep in 1 2; do
for v in fs0 fsb fsr fsbr; do
  GNNX_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_build/ab/libgnnx_hip_$v.so timeout 200 python bench.py --loop-only --steps 20 --warmup 5 --no-cpu-baseline --reps 1 > $O/loop_${v}_$rep.json 2> $O/loop_${v}_$rep.err; show $O/loop_${v}_$rep.json "syn1 $v run $rep"
done; done
for v in fs0 fsbr; do
  for w in syn5 syn4; do GNNX_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_build/ab/libgnnx_hip_$v.so timeout 200 python bench.py --workload $w --loop-only --steps 20 --warmup 5 --no-cpu-baseline --reps 1 > $O/loop_${w}_$v.json 2> /dev/null; show $O/loop_${w}_$v.json "$w $v"; done
  GNNX_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_build/ab/libgnnx_hip_$v.so timeout 300 python bench.py --workload ba100k --targets 2048 --loop-only --steps 5 --warmup 2 --no-cpu-baseline --reps 1 > $O/loop_ba2048_$v.json 2> /dev/null; show $O/loop_ba2048_$v.json "ba100k 2048 $v"
  GNNX_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_build/ab/libgnnx_hip_$v.so timeout 200 python tools/probe_sparse.py 0 2>/dev/null | grep -v amdgpu > $O/timeline_n310_$v.txt
  GNNX_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_build/ab/libgnnx_hip_$v.so timeout 200 python tools/probe_sparse.py 150 2>/dev/null | grep -v amdgpu > $O/timeline_onewave_$v.txt
done
paste -d'|' $O/timeline_n310_fs0.txt $O/timeline_n310_fsbr.txt | cut -c1-400 | tail -12
tail -3 $O/timeline_onewave_fs0.txt; tail -3 $O/timeline_onewave_fsbr.txt
