#!/bin/bash
# round 5, session t: service kernels capped at 32 registers per lane (k_khop, k_count_edges, k_count_edges_large) with the resident kernel at
# 240 / 232 / 224, against the shipped build and session s's 232 build; 300-batch timed regions, two alternations, syn1 and syn5
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5t}; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10"
run() { # variant workload tag
  if [ $1 = shipped ]; then L=""; else L="GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$1.so"; fi
  env $L timeout 300 $B --workload $2 2>/dev/null | tail -1 > $O/bench_$2_$1_$3.json
}
for i in 1 2; do for v in shipped room232b r240s r232s r224s; do run $v syn1 $i; run $v syn5 $i; done; done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), [round(v/1000) for v in e['repetitions']['values']], 'prepare', round(e.get('prepare_ms',0),2), 'khop', round(e.get('khop_ms',0),2), 'plan', round(e.get('plan_pack_route_layout_ms',0),2))" 2>&1 | tail -1; done
