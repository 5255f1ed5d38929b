#!/bin/bash
# round 5, session s: shipped build vs the 232-register / 5 KB-of-LDS-free build with LONGER timed regions (300 batches per repetition instead of
# 50: the 75 ms regions of sessions q / r put syn4 / syn5 inside their own noise), three alternations
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5s}; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10"
run() { # variant workload tag
  if [ $1 = shipped ]; then L=""; else L="GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$1.so"; fi
  env $L timeout 300 $B --workload $2 2>/dev/null | tail -1 > $O/bench_$2_$1_$3.json
}
for i in 1 2 3; do for v in shipped room232b; do run $v syn5 $i; run $v syn4 $i; run $v syn1 $i; done; done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), [round(v/1000) for v in e['repetitions']['values']], 'prepare', round(e.get('prepare_ms',0),2), 'khop', round(e.get('khop_ms',0),2), 'plan', round(e.get('plan_pack_route_layout_ms',0),2))" 2>&1 | tail -1; done
