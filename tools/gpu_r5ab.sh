#!/bin/bash
# round 5, session ab: the dense streaming sample again (four batches in flight for batches with streaming targets), pipeline tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5ab}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "pipeline" > $O/pytest_sub.log 2>&1; tail -1 $O/pytest_sub.log
GNNX_SPARSE_RESIDENT=0 timeout 600 python bench.py --steps 2 --warmup 1 --workload ba100k --targets 1024 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_ba100k_1024targets_dense_streaming.json
timeout 300 python bench.py --workload ba100k --targets 2048 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_ba100k_2048targets.json
for f in $O/r05_bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), r['kernel'][:30], r.get('bound'), round(r['frac'],4), 'd', e['optimisations_in_flight'])" 2>/dev/null; done
