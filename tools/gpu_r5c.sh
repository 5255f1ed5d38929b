#!/bin/bash
# round 5, session c: pair workgroups on the headline, same box, alternating: the round-4 kernel (no pair body, pairing off) vs the new kernel with pairing on
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5c}; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --reps 9"
for i in 1 2 3; do
  GNNX_PAIR_256=0 GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_nopair.so timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_r4kernel_$i.json
  timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_pairon_$i.json
done
GNNX_PAIR_256=0 timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_pairoff_newkernel.json
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), 'wgs', r.get('workgroups'), [round(v/1000) for v in d['end_to_end_stage_ms']['repetitions']['values']], 'one batch', round(d['pcie_inclusive']['batch_total_ms'],2))" 2>&1 | tail -1; done
