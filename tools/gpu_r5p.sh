#!/bin/bash
# round 5, session p: room on a resident workgroup's compute unit for the prepare stage's kernels (mixed kernel capped at 232 / 224 registers
# per lane instead of 256, its LDS pool 2 KB smaller: a 256-thread prepare workgroup with <= 48 registers and 3 KB of LDS then fits BESIDE it)
# - same box, alternating with the shipped build and with the pool-only control
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5p}; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --reps 7"
for i in 1 2; do
  for v in shipped poolonly room232 room224; do
    if [ $v = shipped ]; then L=""; else L="GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so"; fi
    env $L timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_${v}_$i.json
  done
done
for v in shipped room232; do
  if [ $v = shipped ]; then L=""; else L="GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so"; fi
  env $L timeout 400 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ba100k_${v}.json
done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), [round(v/1000) for v in e['repetitions']['values']], 'prepare', round(e.get('prepare_ms',0),2), 'khop', round(e.get('khop_ms',0),2), 'plan', round(e.get('plan_pack_route_layout_ms',0),2), d.get('parity',{}).get('rule','')[:40])" 2>&1 | tail -1; done
