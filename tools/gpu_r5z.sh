#!/bin/bash
# round 5, session z: the new pipeline defaults (two prepare workers, automatic number of optimisations in flight) - pipeline tests, the
# driver's command three times, steady state on syn1 / syn5 / syn4, the 16 384-target set
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5z}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "pipeline or scaling or sharded" > $O/pytest_sub.log 2>&1; tail -2 $O/pytest_sub.log
run() { tag=$1; shift; env "$@" 2>/dev/null | tail -1 > $O/bench_$tag.json; }
for i in 1 2 3; do run syn1_k20_$i A=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline; done
for i in 1 2; do for W in syn1 syn5 syn4; do run ${W}_k300_$i A=1 timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10 --workload $W; done; done
for W in syn5 syn4; do run ${W}_k20 A=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --workload $W; done
run ba100k A=1 timeout 500 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 2 --no-cpu-baseline
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), [round(v/1000) for v in e['repetitions']['values']], 'spread', round(e['repetitions']['spread_pct'],1), 'prepare', round(e.get('prepare_ms',0),2), 'w', e['prepare_workers'], 'd', e['optimisations_in_flight'], 'host core-s', round(e['host_bound_projection']['host_core_seconds_per_step'],4))" 2>&1 | tail -1; done
