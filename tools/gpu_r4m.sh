#!/bin/bash
# round 4, session m: HEAD on hardware (the two k_att commits and the prepare-stage fusion had only run on the emulator), the prepare-stage
# A/B against the round's previous commit (tools/_build/prev: `git archive 491fe07` + its library, built in the container), pipeline knobs.
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
grep -n "FAILED\|^ERROR\|Error" $O/pytest_gpu.log | head -20 >> $O/pytest_gpu_tail.txt
grep -h "every decision identical\|300 epochs from the seeds\|same decisions, beyond\|: tie   id\|ba100k (\|well-conditioned\|beyond 1e-5 (id\|config4 \[full\|config4 (64\|AUC \|\[full\]\|\[early\]\|k_sparse_large vs streaming" $O/pytest_gpu.log | grep -v "^E " | cut -c1-1300 > $O/r04_parity_lines.txt
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1 >> $O/pytest_gpu_tail.txt
Q="--no-parity-gate --no-cpu-baseline --reps 7"
line() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); e=d['end_to_end_stage_ms']
print('$2', 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'prepare', round(e.get('prepare_ms',0),2), 'khop', round(e.get('khop_ms',0),2), 'plan+pack+route', round(e.get('plan_pack_route_ms',0),2), 'layout', round(e.get('edge_layout_ms',0),2), 'rng', round(e.get('host_rng_ms',0),2), [round(v/1e3,1) for v in e['repetitions']['values']], 'single batch', round(d.get('pcie_inclusive',{}).get('batch_total_ms',0),2))" 2>&1 | tail -1; }
[ -e tools/_build/prev/tests/golden ] || ln -sfn $GRAFT_REPO_ROOT/tests/golden tools/_build/prev/tests/golden
for rep in 1 2; do
  timeout 300 python bench.py $Q > $O/bench_new_$rep.json 2> /dev/null; line $O/bench_new_$rep.json "new  run $rep"
  timeout 300 python tools/_build/prev/bench.py $Q > $O/bench_prev_$rep.json 2> /dev/null; line $O/bench_prev_$rep.json "prev run $rep"
done
GNNX_KEEP_256=1 timeout 300 python bench.py $Q > $O/bench_keep256.json 2> /dev/null; line $O/bench_keep256.json "keep256"
GNNX_PIPE_WORKERS=2 timeout 300 python bench.py $Q > $O/bench_w2.json 2> /dev/null; line $O/bench_w2.json "workers 2"
GNNX_PIPE_WORKERS=4 timeout 300 python bench.py $Q > $O/bench_w4.json 2> /dev/null; line $O/bench_w4.json "workers 4"
GNNX_PIPE_DEPTH=3 timeout 300 python bench.py $Q > $O/bench_d3.json 2> /dev/null; line $O/bench_d3.json "depth 3"
GNNX_PIPE_DEPTH=6 GNNX_PIPE_WORKERS=4 timeout 300 python bench.py $Q > $O/bench_d6w4.json 2> /dev/null; line $O/bench_d6w4.json "depth 6 workers 4"
timeout 300 python bench.py > $O/r04_bench_syn1_default_m.json 2> $O/bench_default.err; line $O/r04_bench_syn1_default_m.json "driver command"
timeout 300 python tools/probe_att.py 2>/dev/null | grep -v Warning > $O/r04_method_att_syn1_400targets_m.txt; head -3 $O/r04_method_att_syn1_400targets_m.txt
cat $O/pytest_gpu_tail.txt
