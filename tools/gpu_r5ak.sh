#!/bin/bash
# round 5, session ak: the register cap of the mixed kernel once more with the leaner loop (in-tree = 224: two 32-register service waves per SIMD;
# 232 / 240: one) - steady state and the driver's 20-batch regions, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5ak}; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" 2>/dev/null | tail -1 > $O/bench_$tag.json; }
for i in 1 2 3; do
  for v in intree cap232 cap240; do
    if [ $v = intree ]; then L="A=1"; else L="GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so"; fi
    run syn1_k300_${v}_$i $L timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10
    run syn1_k20_${v}_$i $L timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
  done
done
for v in intree cap240; do
  if [ $v = intree ]; then L="A=1"; else L="GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_$v.so"; fi
  run syn5_k300_${v} $L timeout 300 python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10 --workload syn5
done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), [round(v/1000) for v in e['repetitions']['values']], 'prepare', round(e.get('prepare_ms',0),2))" 2>&1 | tail -1; done
