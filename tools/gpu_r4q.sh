#!/bin/bash
# round 4, session q: which part of the fast head costs parity margin - normalisations (hn), softmax (hs), both (h3), neither (h00)
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'loop ms', round(d['ms_per_step'],4), 'kernel ms', round(r.get('avg_launch_us', 0)/1e3,4), 'frac', round(r['frac'],4), d.get('parity',{}).get('rule','')[:60])" 2>&1 | tail -1; }
for v in h00 hn hs h3; do
  export GNNX_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_build/ab/libgnnx_hip_$v.so
  timeout 200 python bench.py --loop-only --steps 20 --warmup 5 --no-cpu-baseline --reps 1 > $O/loop_${v}.json 2> $O/loop_${v}.err; show $O/loop_${v}.json "syn1 $v"
  timeout 300 python -m pytest tests/test_windowed_parity.py tests/test_decision_parity.py -m gpu -q -s -k "every_target_every_window_gpu or full_horizon_decisions_node" > $O/parity_$v.log 2>&1
  tail -1 $O/parity_$v.log
  grep -h "well-conditioned windows\|every decision identical\|300 epochs from the seeds" $O/parity_$v.log | grep -v "^E  " | cut -c1-330
done
