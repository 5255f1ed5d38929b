#!/bin/bash
# round 4, session b: the whole GPU suite with the logging form, the decision-parity test (every target, every window), bench lines
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
export GNNX_DUMP_WINDOWS=$O/windows
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -h "decisions identical\|tie   \|FAIL\|expansive id\|passed\|failed\|Error" $O/pytest_gpu.log | cut -c1-700 | head -150
timeout 600 python bench.py --steps 20 --warmup 5 --no-parity-gate --no-cpu-baseline > $O/bench_syn1.json 2> $O/bench_syn1.err; echo "bench rc=$?" >> $O/bench_syn1.err
timeout 900 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 1 --no-parity-gate --no-cpu-baseline > $O/bench_ba100k_16384.json 2> $O/bench_ba100k_16384.err
for f in bench_syn1 bench_ba100k_16384; do python -c "
import json;d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]);print('$f value',d['value'],'ms',d['ms_per_step'],'roofline',d['roofline'].get('frac'), 'loop_only', (d.get('loop_only') or {}).get('value'), 'e2e', json.dumps(d.get('end_to_end_stage_ms'))[:600])"; done
