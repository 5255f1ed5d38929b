#!/bin/bash
# round 5, session k: lighter prepare kernels (3 KB k-hop bitmaps for small graphs, a 256-thread analysis kernel for targets of up to 512 nodes) - the
# headline's stage times and rate, config 5 end to end
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5k}; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -x -k "khop or plan_routing or pack or mixed or pipeline" > $O/pytest_sub.log 2>&1; tail -2 $O/pytest_sub.log
B="python bench.py --no-cpu-baseline --reps 7"
for i in 1 2 3; do timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_$i.json; done
timeout 200 $B --workload syn5 2>/dev/null | tail -1 > $O/bench_syn5.json
timeout 500 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ba100k.json
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d['loop_only']['ms_per_step'],3), [round(v/1000) for v in e['repetitions']['values']], 'prepare', round(e['prepare_ms'],2), 'khop', round(e['khop_ms'],2), 'plan', round(e['plan_pack_route_layout_ms'],2), 'one batch', round(d['pcie_inclusive']['batch_total_ms'],2))" 2>&1 | tail -1; done
