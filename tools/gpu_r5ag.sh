#!/bin/bash
# round 5, session ag: the lane's column index declared a per-iteration value as well (GNNX_OPAQUE_LANE=1: 5 spilled registers, 0 / 6 / 0 scratch
# loads per iteration) against session af's winner
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5ag}; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" 2>/dev/null | tail -1 > $O/bench_$tag.json; }
for i in 1 2 3; do
  for v in opq120 opq120_l1; do
    run syn1_k300_${v}_$i GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10
  done
done
for v in opq120 opq120_l1; do run syn5_k300_${v} GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10 --workload syn5; done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), [round(v/1000) for v in e['repetitions']['values']], 'prepare', round(e.get('prepare_ms',0),2))" 2>&1 | tail -1; done
