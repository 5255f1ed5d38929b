#!/bin/bash
# round 5, session ai: independent lane sums taken step by step over all their values (the feature-mask partials' 10 columns, the head's 4 class
# logits) instead of value by value - the compiler had left 2 wait states between the dependent DPP steps of each: -89 s_nop, -103 instructions per
# iteration.  In-tree library = per-iteration addresses only; variant stg = + the staged sums.  Bit-identical results.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5ai}; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" 2>/dev/null | tail -1 > $O/bench_$tag.json; }
for i in 1 2 3; do
  run syn1_k300_intree_$i A=1 timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10
  run syn1_k300_stg_$i GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_stg.so timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10
done
run syn5_k300_intree A=1 timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10 --workload syn5
run syn5_k300_stg GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_stg.so timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10 --workload syn5
GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_stg.so timeout 600 python -m pytest tests -m gpu -q -x -k "pair or mixed_launch or golden_reference_outputs or resumed or logging" > $O/pytest_sub.log 2>&1; tail -1 $O/pytest_sub.log
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), [round(v/1000) for v in e['repetitions']['values']], d.get('parity',{}).get('max_abs_err'))" 2>&1 | tail -1; done
