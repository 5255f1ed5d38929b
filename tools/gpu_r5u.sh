#!/bin/bash
# round 5, session u: the DRIVER's command (--steps 20 --warmup 5: 20-batch timed regions, fill and drain inside) for the shipped build and the
# room builds, three alternations; and two prepare workers instead of three on the 224 build
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5u}; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
run() { # variant tag
  if [ $1 = shipped ]; then L=""; else L="GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$1.so"; fi
  env $L timeout 300 $B 2>/dev/null | tail -1 > $O/bench_syn1_$1_$2.json
}
for i in 1 2 3; do for v in shipped r240s r232s r224s; do run $v $i; done; done
for i in 1 2; do GNNX_PIPE_WORKERS=2 GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_r224s.so timeout 300 $B 2>/dev/null | tail -1 > $O/bench_syn1_r224s_w2_$i.json; done
for i in 1 2; do GNNX_PIPE_WORKERS=4 GNNX_PIPE_DEPTH=6 GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_r224s.so timeout 300 $B 2>/dev/null | tail -1 > $O/bench_syn1_r224s_w4d6_$i.json; done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), [round(v/1000) for v in e['repetitions']['values']], 'spread', round(e['repetitions']['spread_pct'],1), 'prepare', round(e.get('prepare_ms',0),2), 'khop', round(e.get('khop_ms',0),2), 'plan', round(e.get('plan_pack_route_layout_ms',0),2), 'one batch', round(d['pcie_inclusive']['batch_total_ms'],2))" 2>&1 | tail -1; done
