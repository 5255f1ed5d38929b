#!/bin/bash
# round 5, session ah: the per-iteration row / edge addresses in the UNcapped standalone kernels (Tree-Cycles: single-wave class; config 4: graph
# mode): any cost there?
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5ah}; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" 2>/dev/null | tail -1 > $O/bench_$tag.json; }
for i in 1 2; do
  for v in shipped opq120; do
    if [ $v = shipped ]; then L="A=1"; else L="GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so"; fi
    run syn4_k300_${v}_$i $L timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10 --workload syn4
    run config4_${v}_$i $L timeout 300 python bench.py --workload config4 --steps 5 --warmup 2 --no-cpu-baseline
  done
done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']))" 2>&1 | tail -1; done
