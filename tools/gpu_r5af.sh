#!/bin/bash
# round 5, session af: loop invariants the compiler hoists out of the iteration loop and then spills under the 224-register cap (LDS addresses of
# the owned edges, of the row slots, of the head's operands) declared per-iteration values (GNNX_OPAQUE_*): spilled registers 74 -> 9, scratch
# loads per iteration 14 / 65 / 18 -> 0 / 10 / 0 (512-thread / pair / single-wave body).  Loop-only and steady state, alternating.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5af}; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" 2>/dev/null | tail -1 > $O/bench_$tag.json; }
for i in 1 2 3; do
  for v in shipped opq121 opq120 opq121_nocap; do
    if [ $v = shipped ]; then L="A=1"; else L="GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so"; fi
    run syn1_k300_${v}_$i $L timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10
    [ $i = 1 ] && run syn5_k300_${v}_$i $L timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10 --workload syn5
  done
done
timeout 300 python -m pytest tests -m gpu -q -x -k "pair or mixed_launch or golden_reference_outputs_node" > $O/pytest_sub.log 2>&1; tail -1 $O/pytest_sub.log
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), [round(v/1000) for v in e['repetitions']['values']], 'prepare', round(e.get('prepare_ms',0),2))" 2>&1 | tail -1; done
