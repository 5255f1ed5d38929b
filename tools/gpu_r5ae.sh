#!/bin/bash
# round 5, session ae: the automatic confinement to one NUMA node (package import, before torch) - the driver's command four times, steady state twice,
# and GNNX_CPU_AFFINITY=0 as the control
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5ae}; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
for i in 1 2 3 4; do
  timeout 300 $B 2>/dev/null | tail -1 > $O/bench_syn1_auto_$i.json
  GNNX_CPU_AFFINITY=0 timeout 300 $B 2>/dev/null | tail -1 > $O/bench_syn1_free_$i.json
done
for i in 1 2; do timeout 300 python bench.py --steps 300 --warmup 10 --reps 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_syn1_auto_k300_$i.json; done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), [round(v/1000) for v in e['repetitions']['values']], 'spread', round(e['repetitions']['spread_pct'],1), 'prepare', round(e.get('prepare_ms',0),2), e.get('cpu_affinity'))" 2>&1 | tail -1; done
