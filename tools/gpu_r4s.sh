#!/bin/bash
# round 4, session s: the 512-thread class without spills (GNNX_OPAQUE on the packed edge indices: opq = the shipped default), and the
# publish merged into the edge phase for that class too now that it has registers to spare (mp512)
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'loop ms', round(d['ms_per_step'],4), 'kernel ms', round(r.get('avg_launch_us', 0)/1e3,4), 'frac', round(r['frac'],4), d.get('parity',{}).get('rule','')[:60])" 2>&1 | tail -1; }
for rep in 1 2; do
for v in opq mp512; do
  GNNX_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_build/ab/libgnnx_hip_$v.so timeout 200 python bench.py --loop-only --steps 20 --warmup 5 --no-cpu-baseline --reps 1 > $O/loop_${v}_$rep.json 2> $O/loop_${v}_$rep.err; show $O/loop_${v}_$rep.json "syn1 $v run $rep"
done; done
timeout 200 python tools/probe_sparse.py 0 2>/dev/null | grep -v amdgpu > $O/timeline_n310_opq.txt; tail -11 $O/timeline_n310_opq.txt | cut -c1-200
