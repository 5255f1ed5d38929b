#!/bin/bash
# round 5, session q: how much room beside a resident workgroup pays (232 / 216 / 208 registers per lane, 3.8 / 5.3 KB of LDS left), and the
# prepare workers x optimisations in flight sweep again now that the prepare kernels co-reside
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5q}; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --reps 7"
for i in 1 2; do
  for v in room232 room232b room216 room208; do
    GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_${v}_w3_d4_$i.json
  done
done
for wd in "4 4" "4 6" "5 6" "6 8" "3 6"; do
  set -- $wd
  GNNX_PIPE_WORKERS=$1 GNNX_PIPE_DEPTH=$2 GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_room232.so timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_room232_w$1_d$2_1.json
  GNNX_PIPE_WORKERS=$1 GNNX_PIPE_DEPTH=$2 GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_room208.so timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_room208_w$1_d$2_1.json
done
for W in syn4 syn5; do
  timeout 200 $B --workload $W 2>/dev/null | tail -1 > $O/bench_${W}_shipped.json
  GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_room232.so timeout 200 $B --workload $W 2>/dev/null | tail -1 > $O/bench_${W}_room232.json
done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), [round(v/1000) for v in e['repetitions']['values']], 'prepare', round(e.get('prepare_ms',0),2), 'khop', round(e.get('khop_ms',0),2), 'plan', round(e.get('plan_pack_route_layout_ms',0),2), d.get('parity',{}).get('rule','')[:40])" 2>&1 | tail -1; done
