#!/bin/bash
# round 5, session x: the driver's 20-batch regions - prepare workers x optimisations in flight x CPython's GIL hand-over interval
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5x}; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B 2>/dev/null | tail -1 > $O/bench_syn1_$tag.json; }
for i in 1 2; do
  run w2_d4_$i GNNX_PIPE_WORKERS=2
  run w2_d3_$i GNNX_PIPE_WORKERS=2 GNNX_PIPE_DEPTH=3
  run w2_d6_$i GNNX_PIPE_WORKERS=2 GNNX_PIPE_DEPTH=6
  run w2_d4_sw05_$i GNNX_PIPE_WORKERS=2 GNNX_SWITCH_INTERVAL=0.0005
  run w2_d4_sw01_$i GNNX_PIPE_WORKERS=2 GNNX_SWITCH_INTERVAL=0.0001
  run w3_d4_sw05_$i GNNX_PIPE_WORKERS=3 GNNX_SWITCH_INTERVAL=0.0005
  run w1_d4_$i GNNX_PIPE_WORKERS=1
done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), [round(v/1000) for v in e['repetitions']['values']], 'spread', round(e['repetitions']['spread_pct'],1), 'prepare', round(e.get('prepare_ms',0),2), 'khop', round(e.get('khop_ms',0),2), 'plan', round(e.get('plan_pack_route_layout_ms',0),2), 'one batch', round(d['pcie_inclusive']['batch_total_ms'],2))" 2>&1 | tail -1; done
