#!/bin/bash
# round 4, session a: constant-feature form of the sparse resident kernel - the GPU suite, the default bench line, the loop-only
# line with and without the form (GNNX_XCONST=0), the phase timelines of the n = 310 and a one-wave target
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-parity-gate --no-cpu-baseline > $O/bench_syn1.json 2> $O/bench_syn1.err; echo "bench rc=$?" >> $O/bench_syn1.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-parity-gate --no-cpu-baseline --loop-only > $O/bench_syn1_loop.json 2> $O/bench_syn1_loop.err
GNNX_XCONST=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-parity-gate --no-cpu-baseline --loop-only > $O/bench_syn1_loop_general_form.json 2> $O/bench_syn1_loop_general_form.err
for w in syn4 syn5; do timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-parity-gate --no-cpu-baseline --loop-only > $O/bench_${w}_loop.json 2> $O/bench_${w}_loop.err; done
timeout 300 python tools/probe_sparse.py 0 > $O/probe_sparse_0.log 2>&1
timeout 300 python tools/probe_sparse.py 150 > $O/probe_sparse_150.log 2>&1
tail -3 $O/pytest_gpu.log
for f in bench_syn1 bench_syn1_loop bench_syn1_loop_general_form bench_syn4_loop bench_syn5_loop; do python -c "
import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);print('$f value',d['value'],'ms',d['ms_per_step'],'roofline',d['roofline'].get('frac'), 'loop_only', (d.get('loop_only') or {}).get('value'))"; done
tail -11 $O/probe_sparse_0.log; tail -11 $O/probe_sparse_150.log
