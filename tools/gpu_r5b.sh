#!/bin/bash
# round 5, session b: pair workgroups (two 256-thread targets per 512-thread workgroup of the mixed launch) - A/B on the headline and on syn4 / syn5:
# the kernel without the pair body (round-4 behaviour, variant library), the new kernel with pairing switched off, and with it on; + the unrolled latency table
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5b}; mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/micro/chain_latency > $O/r05_chain_latency.txt 2>&1
timeout 300 python -m pytest tests -m gpu -q -x -k "pair or mixed_launch or plan_routing or logging_form or golden_reference_outputs_node" > $O/pytest_pair.log 2>&1; tail -3 $O/pytest_pair.log
B="python bench.py --no-cpu-baseline --reps 7"
for W in syn1 syn5 syn4; do
  GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_nopair.so timeout 200 $B --workload $W 2>/dev/null | tail -1 > $O/bench_${W}_nopairbody.json
  GNNX_PAIR_256=0 timeout 200 $B --workload $W 2>/dev/null | tail -1 > $O/bench_${W}_pairoff.json
  timeout 200 $B --workload $W 2>$O/bench_${W}_pairon.err | tail -1 > $O/bench_${W}_pairon.json
done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), r['bound'], round(r['frac'],3), 'wgs', r.get('workgroups'), [round(v/1000) for v in d['end_to_end_stage_ms']['repetitions']['values']], d.get('parity',{}).get('rule','')[:60])" 2>&1 | tail -2; done
head -9 $O/r05_chain_latency.txt
