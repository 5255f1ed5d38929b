#!/bin/bash
# the -m gpu suite three times in fresh processes + the default bench (flakiness check)
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do timeout 900 python -m pytest tests -m gpu -q --timeout=600 2>&1 | tail -1; done
timeout 600 python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['steps'], d['warmup'], d['parity']['rule'][:60], round(d['cpu_baseline']['value'],2))"
