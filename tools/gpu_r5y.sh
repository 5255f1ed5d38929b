#!/bin/bash
# round 5, session y: two prepare workers with three / four optimisations in flight against the old (3, 4): steady state (300-batch regions) on
# syn1 / syn5 / syn4, the 16 384-target BA-House x100k set, and the driver's 20-batch regions once more
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5y}; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" 2>/dev/null | tail -1 > $O/bench_$tag.json; }
for i in 1 2; do
  for wd in "2 3" "2 4" "3 4"; do
    set -- $wd
    for W in syn1 syn5 syn4; do
      run ${W}_k300_w$1_d$2_$i GNNX_PIPE_WORKERS=$1 GNNX_PIPE_DEPTH=$2 timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10 --workload $W
    done
    run syn1_k20_w$1_d$2_$i GNNX_PIPE_WORKERS=$1 GNNX_PIPE_DEPTH=$2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
  done
done
for wd in "2 3" "3 4"; do
  set -- $wd
  run ba100k_w$1_d$2 GNNX_PIPE_WORKERS=$1 GNNX_PIPE_DEPTH=$2 timeout 500 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 2 --no-cpu-baseline
done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), [round(v/1000) for v in e['repetitions']['values']], 'spread', round(e['repetitions']['spread_pct'],1), 'prepare', round(e.get('prepare_ms',0),2), 'khop', round(e.get('khop_ms',0),2), 'plan', round(e.get('plan_pack_route_layout_ms',0),2), 'host core-s', round(e['host_bound_projection']['host_core_seconds_per_step'],4))" 2>&1 | tail -1; done
