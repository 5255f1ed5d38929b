#!/bin/bash
# round 5, session d: with pair workgroups a batch holds 116 instead of 141 workgroups - is the pipeline now bound by its prepare stage?  workers x depth sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5d}; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --reps 7"
for cfg in "3 4" "4 4" "5 4" "3 6" "4 6" "5 6" "4 8" "6 8" "3 4"; do
  set -- $cfg
  GNNX_PIPE_WORKERS=$1 GNNX_PIPE_DEPTH=$2 timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_w$1_d$2_$RANDOM.json
done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), [round(v/1000) for v in e['repetitions']['values']], 'prepare', round(e['prepare_ms'],2), 'khop', round(e['khop_ms'],2), 'plan', round(e['plan_pack_route_layout_ms'],2), 'rng', round(e['host_rng_ms'],2), 'hostcpu', round(e['host_bound_projection']['host_core_seconds_per_step']*1e3,1))" 2>&1 | tail -1; done
