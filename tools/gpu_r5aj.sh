#!/bin/bash
# round 5, session aj: v + shfl_xor(v, 32 / 16) through gfx950's lane swaps instead of ds_bpermute (variant pl) against the staged-sums build (stg)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5aj}; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" 2>/dev/null | tail -1 > $O/bench_$tag.json; }
for i in 1 2 3; do
  for v in stg pl; do
    run syn1_k300_${v}_$i GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10
  done
done
for v in stg pl; do run syn5_k300_${v} GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10 --workload syn5; done
GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_pl.so timeout 600 python -m pytest tests -m gpu -q -x -k "pair or mixed_launch or golden_reference_outputs or resumed or logging or att" > $O/pytest_sub.log 2>&1; tail -1 $O/pytest_sub.log
for v in stg pl; do GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so timeout 200 python tools/probe_att.py 2>/dev/null | grep "k_att" | tail -1 | sed "s/^/$v: /"; done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), [round(v/1000) for v in e['repetitions']['values']], d.get('parity',{}).get('max_abs_err'))" 2>&1 | tail -1; done
