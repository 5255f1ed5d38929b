#!/bin/bash
# round 5, session g: the feature-mask chain on the last wave (which then owns no edges) instead of wave 0 - same box, alternating with the variant without it
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5g}; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -x -k "pair or mixed_launch or logging_form or golden_reference_outputs_node or resumed_segments" > $O/pytest_sub.log 2>&1; tail -2 $O/pytest_sub.log
B="python bench.py --no-cpu-baseline --reps 7"
for i in 1 2 3; do
  GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_nooffload.so timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_nooffload_$i.json
  timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_offload_$i.json
done
for W in syn5 syn4; do
  GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_nooffload.so timeout 200 $B --workload $W 2>/dev/null | tail -1 > $O/bench_${W}_nooffload.json
  timeout 200 $B --workload $W 2>/dev/null | tail -1 > $O/bench_${W}_offload.json
done
timeout 60 python tools/probe_sparse.py 0 2>/dev/null | grep -v amdgpu > $O/r05_timeline_sparse_resident_syn1_n310.txt
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), 'launch', round(r['avg_launch_us']), r['bound'], round(r['frac'],3), [round(v/1000) for v in d['end_to_end_stage_ms']['repetitions']['values']], d.get('parity',{}).get('rule','')[:50])" 2>&1 | tail -1; done
tail -4 $O/r05_timeline_sparse_resident_syn1_n310.txt | cut -c1-250
