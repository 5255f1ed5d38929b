#!/bin/bash
# round 4, session c: the GPU suite with the algebraic constant-feature form as default, decision parity incl. the full horizon,
# loop-only bench lines per form, timelines, the 16 384-target set end to end, pipeline knobs
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
export GNNX_DUMP_WINDOWS=$O/windows
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -h "every decision identical\|every decision of all\|FAILED\|passed\|failed\| Error" $O/pytest_gpu.log | cut -c1-900 | head -40
grep -h "same decisions, beyond\|: tie   id" $O/pytest_gpu.log | grep -v config4 | cut -c1-400 | head -60
for form in 2 1; do GNNX_XCONST=$form timeout 600 python bench.py --steps 20 --warmup 5 --no-parity-gate --no-cpu-baseline --loop-only > $O/bench_syn1_loop_form$form.json 2> $O/bench_syn1_loop_form$form.err; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-parity-gate --no-cpu-baseline > $O/bench_syn1.json 2> $O/bench_syn1.err; echo "bench rc=$?" >> $O/bench_syn1.err
GNNX_PIPE_WORKERS=3 timeout 600 python bench.py --steps 20 --warmup 5 --no-parity-gate --no-cpu-baseline > $O/bench_syn1_workers3.json 2> $O/bench_syn1_workers3.err
timeout 900 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 1 --reps 3 --no-parity-gate --no-cpu-baseline > $O/bench_ba100k_16384.json 2> $O/bench_ba100k_16384.err
for f in bench_syn1_loop_form2 bench_syn1_loop_form1 bench_syn1 bench_syn1_workers3 bench_ba100k_16384; do python -c "
import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);print('$f value',round(d['value']),'ms',round(d['ms_per_step'],3),'roofline',round(d['roofline'].get('frac'),4), 'loop_only', round((d.get('loop_only') or {}).get('value',0)), 'e2e', json.dumps(d.get('end_to_end_stage_ms'))[:900])"; done
tail -3 $O/bench_ba100k_16384.err | cut -c1-300
timeout 300 python tools/probe_sparse.py 0 > $O/probe_sparse_0.log 2>&1
timeout 300 python tools/probe_sparse.py 150 > $O/probe_sparse_150.log 2>&1
tail -11 $O/probe_sparse_0.log; tail -11 $O/probe_sparse_150.log
